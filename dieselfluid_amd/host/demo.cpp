// demo.cpp -- the reference's own SPH tests, restated against the C++ host mirror
// (dieselfluid.hpp) and run on the GPU through libdslsph.so:
//   TestGPUCompile   model/sph/sph_test.go:8-12      sph.Init(1.0, vector.Vec{}, nil, 16, true)
//   TestOpenCompute  solver/pcisph/pcisph_test.go:10-17  PciMethod.Run in a thread, then "QUIT"
// plus the WCSPH free-fall known answer through solver::WCSPH::Run_ and one
// GPUPredictorCorrector cycle through the compute::ComputeGPU facade.
// Prints one "key value" line per check; tests/test_gpu_host.py parses them.
#include <cstdio>
#include <thread>

#include "dieselfluid.hpp"

using namespace dsl;

int main() {
  try {
    {  // TestGPUCompile: zero-length origin collapses the lattice onto (0,0,0) but Init must return
      sph::SPH s = sph::SPH::Init(1.0f, {}, nullptr, 16, true);
      auto pos = s.Positions();
      float mx = 0.f;
      for (float v : pos) mx = std::fabs(v) > mx ? std::fabs(v) : mx;
      std::printf("TestGPUCompile n %d max_abs_pos %.9g delta %.9g\n", s.N(), mx, s.Delta());
    }
    {  // sph.Init with a proper origin: densities / delta for the parity test
      sph::SPH s = sph::SPH::Init(1.0f, {0.f, 0.f, 0.f}, nullptr, 16, true);
      auto rho = s.Densities();
      auto f = s.Forces();
      double sum = 0;
      for (float r : rho) sum += r;
      std::printf("Init16 n %d delta %.9g rho_mean %.9g rho0 %.9g f1y %.9g\n", s.N(), s.Delta(), sum / rho.size(),
                  s.params().ref_density, f[1]);
    }
    {  // TestOpenCompute: 4096 particles, PCISPH loop in its own thread, quit after a step
      sph::SPH s = sph::SPH::Init(1.0f, {0.f, 0.f, 0.f}, nullptr, 16, true);
      solver::PciMethod pci(&s);
      Chan<std::string> message;
      std::thread t([&] { pci.Run(message); });
      std::string m = message.recv();  // first "SAMPLER_UPDATE" => one step completed
      message.send("QUIT");
      t.join();
      std::printf("TestOpenCompute first_message %s steps %ld\n", m.c_str(), pci.steps());
    }
    {  // WCSPH free fall through Run_ and the THREAD_* channel protocol
      sph::SPH s = sph::SPH::Init(1.0f, {0.f, 0.f, 0.f}, nullptr, 8, false);
      // Init leaves F = gravity + viscous(=0 at rest); the loop adds gravity again (2g)
      solver::WCSPH w(s, 3);
      Chan<int> t;
      std::thread th([&] { w.Run_(t); });
      for (int k = 0; k < 3; ++k) {
        t.send(THREAD_GO);
        t.send(THREAD_GO);
      }
      th.join();
      auto v = s.Velocities();
      auto x = s.Positions();
      std::printf("WCSPHFreeFall steps %ld vy %.9g y0 %.9g\n", w.steps(), v[1], x[1]);
    }
    {  // GPUPredictorCorrector through the ComputeGPU facade, 2 cycles with read-back
      sph::SPH s = sph::SPH::Init(1.0f, {0.f, 0.f, 0.f}, nullptr, 8, true);
      compute::Descriptor d;
      d.Work = {8, 8, 8};
      d.Local = {4, 4, 4};
      compute::ComputeGPU gpu(&d, &s);
      solver::GPUPredictorCorrector pc(&gpu, &s);
      Chan<std::string> msg;
      std::vector<float> positions;
      int refresh = 0;
      std::thread th([&] {
        for (int k = 0; k < 2; ++k)
          if (msg.recv() == "CL_REFRESH") ++refresh;
      });
      std::string err = pc.Run(&msg, &positions, 2);
      th.join();
      std::string bad = gpu.PassFloatBuffer(positions, "no_such_buffer");
      std::printf("GPUPredictorCorrector err '%s' refresh %d npos %zu valid %d bad_buffer_error %d\n", err.c_str(),
                  refresh, positions.size(), gpu.ValidState() ? 1 : 0, bad.empty() ? 0 : 1);
    }
    std::printf("demo ok\n");
    return 0;
  } catch (const std::exception& e) {
    std::printf("demo FAILED: %s\n", e.what());
    return 1;
  }
}
