// dieselfluid.hpp -- C++ host side above the C ABI (include/dslsph.h), mirroring the
// reference's Go interfaces for the SPH hot path one to one: same type and method names,
// same argument meaning, same error behaviour.  The reference is Go; this image has no
// Go toolchain, so the host layer that a Go maintainer would write in Go (see
// bindings/go/dslsph and INTEGRATION.md) is provided in C++ where the tests can build and
// run it.  Nothing here computes particle physics on the CPU: every pass is a call into
// libdslsph.so.  Host-side work is limited to what the reference itself does on the host
// before handing buffers to the device: the initial lattice (geom/grid/point-grid.go) and
// the PCISPH delta scalar (model/sph/fluid.go:221-277).
//
// Mirrors (file:line in the dieselfluid repository):
//   dsl::sph::SPH                 model/sph/fluid.go:23-277
//   dsl::solver::SPHMethod        solver/method.go:3-6
//   dsl::solver::WCSPH            solver/wcsph/wcsph.go:9-75
//   dsl::solver::PciMethod        solver/pcisph/pcisph_darwin.go:11-118
//   dsl::solver::GPUPredictorCorrector  solver/pcisph/pcisph_gpu_darwin.go:22-286
//   dsl::compute::Descriptor / ComputeGPU   compute/compute.go:9-53, compute/gpu/gpu.go:20-425
//   dsl::Chan<T>                  Go unbuffered channel used by the drivers
#pragma once

#include <array>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/dslsph.h"

namespace dsl {

// model/model.go:11-15 thread message enums
constexpr int THREAD_WAIT = 100;
constexpr int THREAD_GO = 101;
constexpr int THREAD_ERR = 102;
constexpr int THREAD_DONE = 103;
constexpr int SPH_THREAD_WAITING = 104;

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// Minimal stand-in for a Go channel: blocking send/recv plus the non-blocking forms the
// PCISPH driver uses in its select{} statements (pcisph_darwin.go:103-116).
template <class T>
class Chan {
 public:
  void send(T v) {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return !slot_.has_value(); });
    slot_ = std::move(v);
    cv_.notify_all();
  }
  T recv() {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return slot_.has_value(); });
    T v = std::move(*slot_);
    slot_.reset();
    cv_.notify_all();
    return v;
  }
  bool try_send(T v) {
    std::lock_guard<std::mutex> lk(m_);
    if (slot_.has_value()) return false;
    slot_ = std::move(v);
    cv_.notify_all();
    return true;
  }
  bool try_recv(T& out) {
    std::lock_guard<std::mutex> lk(m_);
    if (!slot_.has_value()) return false;
    out = std::move(*slot_);
    slot_.reset();
    cv_.notify_all();
    return true;
  }

 private:
  std::mutex m_;
  std::condition_variable cv_;
  std::optional<T> slot_;
};

// ---------------------------------------------------------------------------------------
// model/sph
// ---------------------------------------------------------------------------------------
namespace sph {

constexpr float VISCOSITY_WATER = 1.3059f;  // fluid.go:18
constexpr float CACHE_L = 0.8f;             // fluid.go:19

// geom/grid/point-grid.go:26-28,39-40,53-63 + model/field/sph_field.go:87-108.
// origin_len != 3 reproduces V.Add's length-mismatch result (vector.go:171-173).
inline std::vector<float> LatticePositions(int n3, const float* origin, int origin_len) {
  std::vector<float> pos((size_t)n3 * n3 * n3 * 3);
  float minb[3], step[3];
  const float dim = (float)n3;
  for (int a = 0; a < 3; ++a) {
    const float m = -1.0f * 1.0f;
    minb[a] = origin_len == 3 ? m + origin[a] : 0.0f;
  }
  const float inv = 1.0f / dim;
  for (int a = 0; a < 3; ++a) {
    const float t = minb[a] * -2.0f;
    step[a] = inv * t;
  }
  for (int i = 0; i < n3; ++i)
    for (int j = 0; j < n3; ++j)
      for (int k = 0; k < n3; ++k) {
        const int id = k + n3 * (i * n3 + j);
        const float ijk[3] = {(float)i, (float)j, (float)k};
        for (int a = 0; a < 3; ++a) {
          const float sv = step[a] * ijk[a];
          pos[(size_t)3 * id + a] = minb[a] + sv;
        }
      }
  return pos;
}

}  // namespace sph

// geom/mesh/mesh.go: the part of Mesh the SPH path touches
namespace mesh {
struct Mesh {
  std::vector<std::array<float, 3>> Vertexes;
  // mesh.go:60-76: one particle per vertex (`density` is unused there too); the bound check
  // `x < len(particle_list)-3` leaves the LAST vertex's particle at the origin
  std::vector<float> GenerateBoundaryParticles(float /*density*/) const {
    const int len = (int)Vertexes.size() * 3;
    std::vector<float> particle_list((size_t)len, 0.0f);
    for (int index = 0; index < (int)Vertexes.size(); ++index) {
      const int x = index * 3;
      if (x < len - 3) {
        particle_list[(size_t)x] = Vertexes[(size_t)index][0];
        particle_list[(size_t)x + 1] = Vertexes[(size_t)index][1];
        particle_list[(size_t)x + 2] = Vertexes[(size_t)index][2];
      }
    }
    return particle_list;
  }
};
}  // namespace mesh

namespace sph {

class SPH {
 public:
  // sph.Init(scl, origin, colliders, n3, pci)  fluid.go:41-88.  `colliders` is accepted and
  // ignored exactly as the reference does (BoundaryParticles is commented out, fluid.go:70);
  // `boundary_capacity` reserves device slots for a later BoundaryParticles() call.
  static SPH Init(float scl, const std::vector<float>& origin, const std::vector<mesh::Mesh*>* colliders, int n3, bool pci,
                  int device = 0, int math_mode = DSL_MATH_EXACT, int boundary_capacity = 0) {
    (void)scl;
    (void)colliders;
    SPH core;
    if (dsl_params_reference(&core.prm_, n3) != DSL_OK) throw Error(dsl_last_error(nullptr));
    core.prm_.math_mode = math_mode;
    if (boundary_capacity > 0) core.prm_.capacity = core.prm_.n_particles + boundary_capacity;
    core.particles_ = core.prm_.n_particles;
    core.cache_life_ = CACHE_L;
    core.mu_ = VISCOSITY_WATER;
    if (dsl_create(&core.prm_, device, &core.h_) != DSL_OK) throw Error(dsl_last_error(nullptr));
    std::vector<float> pos = LatticePositions(n3, origin.data(), (int)origin.size());
    core.ck(dsl_upload(core.h_, DSL_BUF_POSITIONS, pos.data(), pos.size()));  // AlignWithGrid :71
    core.NN();                                                                 // UpdateSampler :72
    core.DensityAll();                                                         // :73
    const float g[3] = {0.0f, -9.81f * core.prm_.mass, 0.0f};
    core.ExternalAll(g);                                                       // :74
    core.ViscousAll();                                                         // :75
    core.CFL();                                                                // :76
    if (pci) {                                                                 // :78-82
      if (core.pcidelta() == 0.0f) core.delta_ = core.prm_.h;
      core.prm_.delta = core.delta_;
      core.ck(dsl_set_params(core.h_, &core.prm_));
    }
    return core;
  }

  SPH() = default;
  SPH(const SPH&) = delete;
  SPH& operator=(const SPH&) = delete;
  SPH(SPH&& o) noexcept { *this = std::move(o); }
  SPH& operator=(SPH&& o) noexcept {
    if (this != &o) {
      release();
      std::memcpy(&prm_, &o.prm_, sizeof(prm_));
      h_ = o.h_;
      o.h_ = nullptr;
      time_ = o.time_;
      cache_life_ = o.cache_life_;
      mu_ = o.mu_;
      delta_ = o.delta_;
      particles_ = o.particles_;
      boundary_ = o.boundary_;
    }
    return *this;
  }
  ~SPH() { release(); }

  // SPHField.BoundaryParticles (model/field/sph_field.go:75-85): every collider's vertex particles go
  // to ParticleArray.AddBoundaryParticles (particle_array.go:123-128); returns the last collider's list
  std::vector<float> BoundaryParticles(const std::vector<mesh::Mesh*>& colliders) {
    std::vector<float> colliderPositions;
    for (mesh::Mesh* m : colliders) {
      colliderPositions = m->GenerateBoundaryParticles(2.0f);
      if (!colliderPositions.empty()) {
        ck(dsl_add_boundary_particles(h_, colliderPositions.data(), colliderPositions.size()));
        ck(dsl_get_params(h_, &prm_));  // n_boundary has grown: dsl_set_params wants the current count back
      }
      boundary_ += (int)colliderPositions.size() / 3;
    }
    return colliderPositions;
  }
  int Total() const { return particles_ + boundary_; }  // particle_array.go:134-136

  int N() const { return particles_; }                 // fluid.go:106-108
  void NN() { ck(dsl_build_neighbours(h_)); }          // fluid.go:100-102
  float CFL() {                                        // fluid.go:111-114
    time_ = 0.01f;
    return time_;
  }
  float Time() { return CFL(); }                       // fluid.go:199-202
  float Viscosity() const { return mu_; }
  void SetViscosity(float x) {
    mu_ = x;
    prm_.mu = x;
    ck(dsl_set_params(h_, &prm_));
  }
  float Delta() const { return delta_; }               // fluid.go:204
  float MaxV() {                                       // fluid.go:206
    dsl_stats st;
    ck(dsl_get_stats(h_, &st));
    return st.max_vel;
  }
  void DensityAll() { ck(dsl_density_pass(h_)); }                    // fluid.go:127-131
  int PressureAll() {                                                // fluid.go:134-142
    ck(dsl_pressure_pass(h_));
    return 0;  // SPH_VALID
  }
  void ViscousAll() { ck(dsl_viscous_pass(h_)); }                    // fluid.go:146-152
  void ExternalAll(const float f[3]) { ck(dsl_external_pass(h_, f)); }  // fluid.go:155-161
  void GradientPressureForce() { ck(dsl_gradient_pressure_pass(h_)); }  // fluid.go:164-172
  void Update() { ck(dsl_update_pass(h_)); }                         // fluid.go:175-197
  float CacheIncr() {                                                // fluid.go:208-215
    cache_life_ *= cache_life_;
    if (cache_life_ < 0.1f) {
      cache_life_ = CACHE_L;
      NN();
    }
    return cache_life_;
  }

  // ParticleArray accessors (model/particle_array.go:39-54): blocking device reads in the
  // reference's host layout (xyz interleaved).
  std::vector<float> Positions() {  // Total() particles (particle_array.go:21)
    std::vector<float> out((size_t)Total() * 3);
    ck(dsl_download(h_, DSL_BUF_POSITIONS, out.data(), out.size()));
    return out;
  }
  std::vector<float> Velocities() { return read(DSL_BUF_VELOCITIES, 3); }
  std::vector<float> Forces() { return read(DSL_BUF_FORCES, 3); }
  std::vector<float> Densities() { return read(DSL_BUF_DENSITIES, 1); }
  std::vector<float> Pressures() { return read(DSL_BUF_PRESSURES, 1); }

  dsl_handle* handle() { return h_; }
  dsl_params& params() { return prm_; }
  void ck(int rc) const {
    if (rc != DSL_OK) throw Error(dsl_last_error(h_));
  }

  // fluid.go:221-277 pcidelta/computeBeta -- a host-side scalar in the reference too (it
  // is passed to the device in the "floats" block, pcisph_gpu_darwin.go:61).
  float pcidelta() {
    const int n3 = 8, particles = 512;
    const float origin[3] = {0.f, 0.f, 0.f};
    const std::vector<float> pos = LatticePositions(n3, origin, 3);
    const float h = 1.0f;
    const float B = -45.0f / ((float)3.141592653589 * (h * h * h * h));  // std_kernel.go:27
    float denom = 0.0f, d1[3] = {0.f, 0.f, 0.f}, d2 = 0.0f;
    const int mid = particles / 2;
    int tracking = 0;
    auto mag = [](const float* v) {
      float s = 0.0f;
      for (int i = 0; i < 3; ++i) s += v[i] * v[i];
      return (float)std::sqrt((double)s);
    };
    for (int i = 0; i < particles; ++i) {
      int x = mid + tracking;
      if (i % 2 != 0) {
        x = mid - tracking;
        tracking++;
      }
      if (x < 0 || x > particles) break;
      float p[3] = {0.f, 0.f, 0.f};
      if (x < particles) std::memcpy(p, &pos[(size_t)3 * x], sizeof(p));
      const float m = mag(p);
      const float dist2 = m * m;
      if (dist2 < h * h) {
        const float dist = mag(p);
        float dir[3] = {0.f, 0.f, 0.f};
        if (dist > 0.0f) {
          const float inv = 1.0f / dist;
          for (int a = 0; a < 3; ++a) dir[a] = p[a] * inv;
        }
        float o1 = 0.0f;  // O1D, std_kernel.go:54-60
        if (!(dist >= h)) {
          const float q = 1.0f - dist / h;
          const float bq = B * q;
          o1 = bq * q;
        }
        const float s = -o1;
        float g[3];
        for (int a = 0; a < 3; ++a) g[a] = dir[a] * s;
        for (int a = 0; a < 3; ++a) d1[a] = d1[a] + g[a];
        const float t0 = g[0] * g[0], t1 = g[1] * g[1], t2 = g[2] * g[2];
        const float dd = (t0 + t1) + t2;
        d2 += dd;
      }
    }
    {
      const float t0 = d1[0] * d1[0], t1 = d1[1] * d1[1], t2 = d1[2] * d1[2];
      const float dd = (t0 + t1) + t2;
      denom += -dd - d2;
    }
    if (denom != 0.0f) {
      const float t2 = time_ * time_;
      const float m2 = prm_.mass * prm_.mass;
      const float r2 = prm_.ref_density * prm_.ref_density;
      const float beta = t2 * m2 * (2.0f / r2);
      delta_ = -1.0f / (beta * denom);
      return delta_;
    }
    return 0.0f;
  }

 private:
  std::vector<float> read(int buf, int comps) {
    std::vector<float> out((size_t)particles_ * comps);
    ck(dsl_download(h_, buf, out.data(), out.size()));
    return out;
  }
  void release() {
    if (h_) dsl_destroy(h_);
    h_ = nullptr;
  }
  dsl_params prm_{};
  dsl_handle* h_ = nullptr;
  float time_ = 0.0f, cache_life_ = CACHE_L, mu_ = VISCOSITY_WATER, delta_ = 0.0f;
  int particles_ = 0, boundary_ = 0;
};

}  // namespace sph

// ---------------------------------------------------------------------------------------
// solver
// ---------------------------------------------------------------------------------------
namespace solver {

// solver/method.go:3-6
struct SPHMethod {
  virtual ~SPHMethod() = default;
  virtual void Run() = 0;
  virtual void Run_(Chan<int>& t) = 0;
};

// solver/wcsph/wcsph.go:9-75.  The reference's loops never terminate; `max_steps` (0 =
// forever) exists so tests can run them.
class WCSPH : public SPHMethod {
 public:
  explicit WCSPH(sph::SPH& core, long max_steps = 0) : core_(core), max_steps_(max_steps) {}
  void Run() override {  // wcsph.go:14-26: DensityAll, ExternalAll, PressureAll, Update, CFL
    for (long s = 0; max_steps_ == 0 || s < max_steps_; ++s) step();
  }
  void Run_(Chan<int>& t) override {  // wcsph.go:35-75
    bool sync = true;
    for (long s = 0; max_steps_ == 0 || s < max_steps_; ++s) {
      if (sync) {
        step();
        const int status = t.recv();
        if (status == THREAD_WAIT) {
          sync = false;
          t.send(SPH_THREAD_WAITING);
          if (t.recv() == THREAD_GO) sync = true;
        }
        if (status == THREAD_GO) sync = true;
        if (status == THREAD_DONE) return;  // addition: a way out for tests
      }
      if (t.recv() == THREAD_GO) sync = true;
    }
  }
  long steps() const { return steps_; }

 private:
  void step() {
    core_.ck(dsl_wcsph_step(core_.handle(), 1));
    core_.CFL();
    ++steps_;
  }
  sph::SPH& core_;
  long max_steps_, steps_ = 0;
};

// solver/pcisph/pcisph_darwin.go:11-118
class PciMethod {
 public:
  explicit PciMethod(sph::SPH* sys) : system_(sys) {}
  // Run(message chan string, hasGL bool, mRender *render.RenderSystem): the GL arguments
  // belong to the renderer (out of scope) and are dropped.
  void Run(Chan<std::string>& message) {
    system_->ck(dsl_pcisph_begin(system_->handle()));  // :28-41
    bool done = false;
    while (!done) {
      system_->ck(dsl_pcisph_step(system_->handle(), 1));  // :43-101
      ++steps_;
      std::string msg;
      if (message.try_recv(msg) && msg == "QUIT") done = true;  // :103-110
      message.try_send("SAMPLER_UPDATE");                        // :112-116
    }
  }
  long steps() const { return steps_; }

 private:
  sph::SPH* system_;
  long steps_ = 0;
};

}  // namespace solver

// ---------------------------------------------------------------------------------------
// compute / compute/gpu
// ---------------------------------------------------------------------------------------
namespace compute {

// compute/compute.go:9-13
struct Descriptor {
  std::vector<int> Work, Local;
  int Size = 0;
};

// compute/gpu/gpu.go:20-425.  Named buffers map onto the engine's device arrays; the
// kernel names are the reference's (pcisph_gpu_darwin.go:133-139) and select built-in HIP
// kernels instead of compiling OpenCL sources.  Methods return an error string (empty =
// nil) where the Go ones return `error`, and bool where they return bool.
class ComputeGPU {
 public:
  ComputeGPU(Descriptor* desc, sph::SPH* system) : desc_(desc), sys_(system) {}
  std::string RegisterBuffer(int bytes_size, int /*t*/, const std::string& name) {  // gpu.go:314-321
    const int id = buffer_id(name);
    if (id == -2) {
      aux_[name] = bytes_size;  // "sizes","floats","sampler","vecs": parameter blocks, no device array
      return "";
    }
    if (id < 0) return "buffer [" + name + "] is not a buffer of this engine";
    registered_[name] = bytes_size;
    return "";
  }
  std::string PassFloatBuffer(const std::vector<float>& cpu, const std::string& name) {  // gpu.go:343-352
    if (aux_.count(name)) return "";
    if (!registered_.count(name)) return "buffer [" + name + "] not registered";  // gpu.go:305-310
    if (dsl_upload(sys_->handle(), buffer_id(name), cpu.data(), cpu.size()) != DSL_OK)
      return dsl_last_error(sys_->handle());
    return "";
  }
  std::string ReadFloatBuffer(std::vector<float>& cpu, const std::string& name) {  // gpu.go:332-341
    if (!registered_.count(name)) return "buffer [" + name + "] not registered";
    if (dsl_download(sys_->handle(), buffer_id(name), cpu.data(), cpu.size()) != DSL_OK)
      return dsl_last_error(sys_->handle());
    return "";
  }
  bool RegisterKernel(const std::string& name) {  // gpu.go:231-250
    const bool ok = name == "compute_density" || name == "predict_correct";
    if (ok) kernels_[name] = true;
    log_ += ok ? "kernel " + name + " registered\n" : "kernel " + name + " unknown\n";
    return ok;
  }
  std::string AddSourceFile(const std::string&) { return ""; }    // gpu.go:257-271: nothing to compile
  std::string BuildProgram(const std::string&) { return ""; }     // gpu.go:194-229
  std::string Queue(const std::string& name) {                    // gpu.go:286-296
    if (!kernels_.count(name)) return "kernel [" + name + "] not registered";
    pending_ = name;
    return "";
  }
  // ---- the rest of compute.GPUCompute (compute/compute.go:26-53) ----
  bool Setup(bool /*gl_interop*/) { return HasDeviceContext(); }   // compute.go:29: the device context is the handle
  // compute.go:33 Run(x chan int): runs the kernel named by the last Queue() -- the reference's two fused
  // device kernels are phases of the PCISPH step here (pci_density.c:12-23 = NN, density, viscosity;
  // pci_predict.c:9-27 = the correction loop + integrate) -- then reports THREAD_DONE / THREAD_ERR
  void Run(Chan<int>& x) {
    int rc = DSL_OK;
    dsl_handle* h = sys_->handle();
    if (pending_ == "compute_density") {
      rc = dsl_pcisph_begin(h);  // (re-copies the predictor state only the first time it is needed)
      if (rc == DSL_OK) rc = dsl_pcisph_phase(h, DSL_PCI_BEGIN_STEP);
    } else if (pending_ == "predict_correct") {
      for (int it = 0; rc == DSL_OK && it < sys_->params().pci_max_iters; ++it) {
        rc = dsl_pcisph_phase(h, DSL_PCI_ITERATE);
        if (rc == DSL_OK) rc = dsl_pcisph_phase(h, DSL_PCI_CHECK);
      }
      if (rc == DSL_OK) rc = dsl_pcisph_phase(h, DSL_PCI_END_STEP);
    } else {
      rc = DSL_ERR_INVALID;
    }
    if (rc == DSL_OK) rc = dsl_sync(h);
    x.send(rc == DSL_OK ? THREAD_DONE : THREAD_ERR);
  }
  // gpu.go:354-378.  "sizes" = {N, Nboundary, buckets, bucket_size} (pcisph_gpu_darwin.go:60): checked
  // against the engine's parameters instead of being dropped; "sampler" = the host's flattened LSH
  // table (lsh.go:70-80): the engine builds its own neighbour table, the upload is accepted and ignored.
  std::string PassIntBuffer(const std::vector<int>& cpu, const std::string& name) {
    if (!aux_.count(name) && !registered_.count(name)) return "buffer [" + name + "] not registered";
    if (name == "sizes") {
      const dsl_params& p = sys_->params();
      if (cpu.size() < 2 || cpu[0] != p.n_particles || cpu[1] != p.n_boundary)
        return "sizes block {N, Nboundary, ..} does not match the engine's parameters";
    }
    log_ += "Passed Integer Buffer " + name + "\n";
    return "";
  }
  std::string ReadIntBuffer(std::vector<int>& cpu, const std::string& name) {
    if (!aux_.count(name) && !registered_.count(name)) return "buffer [" + name + "] not registered";
    const dsl_params& p = sys_->params();
    if (name == "sizes") {
      const int v[4] = {p.n_particles, p.n_boundary, p.lsh_buckets, p.lsh_bucket_size};
      for (size_t k = 0; k < cpu.size() && k < 4; ++k) cpu[k] = v[k];
      return "";
    }
    if (name == "sampler") {  // HashSampler.GetData1D (DSL_NEIGH_LSH_REF handles only)
      std::vector<int32_t> t(cpu.size());
      if (dsl_lsh_download_table(sys_->handle(), t.data(), t.size()) != DSL_OK) return dsl_last_error(sys_->handle());
      for (size_t k = 0; k < cpu.size(); ++k) cpu[k] = t[k];
      return "";
    }
    return "buffer [" + name + "] holds no integers";
  }
  // gpu.go:378-390 PassLayoutBuffer for the two parameter blocks of pcisph_gpu_darwin.go:60-61
  std::string PassLayoutBuffer(const void* data, int bytes, const std::string& name) {
    if (!aux_.count(name)) return "buffer [" + name + "] not registered";
    const dsl_params& p = sys_->params();
    if (name == "floats" && bytes >= 20) {  // {dt, mass, delta, maxVel, h}
      const float* f = static_cast<const float*>(data);
      if (f[0] != p.dt || f[1] != p.mass || f[4] != p.h) return "floats block {dt, mass, delta, maxVel, h} does not match the engine's parameters";
    }
    if (name == "sizes" && bytes >= 8) {
      const int32_t* v = static_cast<const int32_t*>(data);
      if (v[0] != p.n_particles || v[1] != p.n_boundary) return "sizes block {N, Nboundary, ..} does not match the engine's parameters";
    }
    log_ += "Passed Layout Buffer " + name + "\n";
    return "";
  }
  std::string AddSourceString(const std::string&) { return ""; }  // compute.go:46: nothing to compile
  // gpu.go:323-330 RegisterGLBuffer shares a GL buffer with the device queue so that the renderer draws what
  // the solver wrote without a read-back.  The counterpart here is the other way round: the consumer is
  // handed the live device arrays (dsl_device_pointers); the GL id is recorded, nothing is mapped.
  std::string RegisterGLBuffer(uint32_t gl_buffer_id, int size, const std::string& name) {
    if (buffer_id(name) != DSL_BUF_POSITIONS) return "only the positions buffer has a render hand-off";
    registered_[name] = size;
    log_ += "RegisterGLBuffer() - buffer " + name + " (GL id " + std::to_string(gl_buffer_id) +
            "): read the device arrays through DevicePointers()\n";
    return "";
  }
  // the live SoA device arrays of the positions (x, y, z), slot -> particle id map, slot count
  std::string DevicePointers(const float* xyz[3], const int32_t** ids, int* n) {
    if (dsl_device_pointers(sys_->handle(), DSL_BUF_POSITIONS, xyz, ids, n) != DSL_OK) return dsl_last_error(sys_->handle());
    return "";
  }
  void Set(const Descriptor& d) { *desc_ = d; }     // gpu.go:298-300
  Descriptor Get() const { return *desc_; }         // gpu.go:301-303
  bool HasDeviceContext() const { return sys_ && sys_->handle(); }  // gpu.go:392-398
  bool ValidState() const { return HasDeviceContext(); }            // gpu.go:400-405
  const std::string& Log() const { return log_; }                   // gpu.go:421-425
  sph::SPH* system() { return sys_; }

 private:
  static int buffer_id(const std::string& n) {  // names of pcisph_gpu_darwin.go:67-76
    if (n == "positions") return DSL_BUF_POSITIONS;
    if (n == "velocities") return DSL_BUF_VELOCITIES;
    if (n == "forces") return DSL_BUF_FORCES;
    if (n == "densities") return DSL_BUF_DENSITIES;
    if (n == "pressures") return DSL_BUF_PRESSURES;
    if (n == "temps") return DSL_BUF_PCI_POSITIONS;
    if (n == "sizes" || n == "floats" || n == "sampler" || n == "vecs") return -2;
    return -1;
  }
  Descriptor* desc_;
  sph::SPH* sys_;
  std::map<std::string, int> registered_, aux_;
  std::map<std::string, bool> kernels_;
  std::string pending_, log_;
};

}  // namespace compute

namespace solver {

// solver/pcisph/pcisph_gpu_darwin.go:22-286
class GPUPredictorCorrector {
 public:
  // New_GPUPredictorCorrector(computeGPU, sph, opencl, gl_position_buffer) :36-236
  GPUPredictorCorrector(compute::ComputeGPU* gpu, sph::SPH* system) : gpu_(gpu), system_(system) {
    const int n = system_->N();
    const char* names[] = {"positions", "velocities", "forces", "densities", "pressures"};
    const int bytes[] = {n * 12, n * 12, n * 12, n * 4, n * 4};
    for (int k = 0; k < 5; ++k) must(gpu_->RegisterBuffer(bytes[k], 0, names[k]));  // :67-71
    must(gpu_->RegisterBuffer(4 * 4, 0, "sizes"));                                   // :72
    must(gpu_->RegisterBuffer(5 * 4, 0, "floats"));                                  // :73
    must(gpu_->RegisterBuffer(n * 7 * 4, 0, "temps"));                               // :76
    if (!gpu_->RegisterKernel("compute_density")) throw Error("Register kernel compute density failed");   // :133
    if (!gpu_->RegisterKernel("predict_correct")) throw Error("Register kernel predict_correct failed");  // :137
    system_->ck(dsl_pcisph_begin(system_->handle()));
  }
  // MemoryRequirements() :238-246 (the reference's formula, units as printed there)
  std::string MemoryRequirements() const {
    const int particles = system_->N(), boundary = 0, total = particles;
    const double kb = (double)((total * 4 * 3) + (particles * 4 * 3 * 2) + (2 * particles * 4) / 1024);
    char buf[160];
    std::snprintf(buf, sizeof(buf), "Fluid GPU PCI (Fluid Particles[%d]  Boundary[%d]\nAllocated %.2fkB (%.2fMB)\n\n",
                  particles, boundary, kb, kb * 0.001);
    return buf;
  }
  // Run(message *chan string) error :249-286.  One cycle = both reference kernels
  // (compute_density, predict_correct) = one dsl_pcisph_step, then the blocking position
  // read-back and "CL_REFRESH".  `max_cycles` (0 = forever) lets tests stop the loop.
  std::string Run(Chan<std::string>* message, std::vector<float>* positions, long max_cycles = 0) {
    for (long c = 0; max_cycles == 0 || c < max_cycles; ++c) {
      system_->CFL();        // :253
      system_->CacheIncr();  // :254 (NN every 4th call; the engine also rebuilds every step)
      if (dsl_pcisph_step(system_->handle(), 1) != DSL_OK) return dsl_last_error(system_->handle());  // :256-274
      if (positions) {
        positions->resize((size_t)system_->N() * 3);
        if (dsl_download(system_->handle(), DSL_BUF_POSITIONS, positions->data(), positions->size()) != DSL_OK)
          return dsl_last_error(system_->handle());                                                   // :276-277
      }
      if (message) message->send("CL_REFRESH");                                                       // :279
    }
    return "";
  }

 private:
  static void must(const std::string& e) {
    if (!e.empty()) throw Error(e);
  }
  compute::ComputeGPU* gpu_;
  sph::SPH* system_;
};

}  // namespace solver
}  // namespace dsl
