// Package dslsph is the cgo binding of libdslsph.so (include/dslsph.h), the MI355X
// SPH particle-step engine, shaped to drop in where dieselfluid's OpenCL path sits:
//
//	compute/gpu.ComputeGPU                  -> dslsph.ComputeGPU   (named float buffers)
//	solver/pcisph.GPUPredictorCorrector     -> dslsph.GPUPredictorCorrector
//	solver.SPHMethod (Run / Run_)           -> dslsph.WCSPH, dslsph.PCISPH
//
// NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no Go toolchain (SURVEY.md
// section 0.2).  The file is deliberately thin and mechanical -- every method is one C call
// -- and the same call sequences are exercised from C++ (dieselfluid_amd/host) and Python
// (tests/) against the same library.  Build inside dieselfluid with
//
//	CGO_CFLAGS="-I<repo>/include" CGO_LDFLAGS="-L<repo>/dieselfluid_amd/lib -ldslsph" go build ./...
//
// cgo rules honoured: Go memory is only passed for the duration of a call (dsl_upload /
// dsl_download copy synchronously and retain nothing); the handle is an opaque C pointer;
// every library call re-selects its HIP device because goroutines migrate between OS
// threads.
package dslsph

/*
#cgo LDFLAGS: -ldslsph
#include <stdlib.h>
#include "dslsph.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"unsafe"
)

// Buffer names registered by New_GPUPredictorCorrector (solver/pcisph/pcisph_gpu_darwin.go:67-76).
var bufferIDs = map[string]C.int{
	"positions":  C.DSL_BUF_POSITIONS,
	"velocities": C.DSL_BUF_VELOCITIES,
	"forces":     C.DSL_BUF_FORCES,
	"densities":  C.DSL_BUF_DENSITIES,
	"pressures":  C.DSL_BUF_PRESSURES,
	"temps":      C.DSL_BUF_PCI_POSITIONS,
}

// parameter-block buffers of the reference that have no device array here
var auxBuffers = map[string]bool{"sizes": true, "floats": true, "sampler": true, "vecs": true}

// Engine owns one dsl_handle.
type Engine struct {
	h      *C.dsl_handle
	Params C.dsl_params
}

func lastError(h *C.dsl_handle) error { return errors.New(C.GoString(C.dsl_last_error(h))) }

// ReferenceParams returns sph.Init's constants for an n3^3 system (model/sph/fluid.go:41-88).
func ReferenceParams(n3 int) (C.dsl_params, error) {
	var p C.dsl_params
	if rc := C.dsl_params_reference(&p, C.int(n3)); rc != 0 {
		return p, lastError(nil)
	}
	return p, nil
}

// NewEngine replaces gpu.InitOpenCL + gpu.New_ComputeGPU (compute/gpu/gpu.go:45-119).
func NewEngine(p C.dsl_params, device int) (*Engine, error) {
	e := &Engine{Params: p}
	if rc := C.dsl_create(&e.Params, C.int(device), &e.h); rc != 0 {
		return nil, lastError(nil)
	}
	return e, nil
}

func (e *Engine) Close() {
	if e.h != nil {
		C.dsl_destroy(e.h)
		e.h = nil
	}
}

func (e *Engine) ck(rc C.int) error {
	if rc != 0 {
		return lastError(e.h)
	}
	return nil
}

// Upload / Download: ComputeGPU.PassFloatBuffer / ReadFloatBuffer (gpu.go:343-352,332-341).
func (e *Engine) Upload(buffer C.int, data []float32) error {
	if len(data) == 0 {
		return nil
	}
	return e.ck(C.dsl_upload(e.h, buffer, (*C.float)(unsafe.Pointer(&data[0])), C.size_t(len(data))))
}

func (e *Engine) Download(buffer C.int, data []float32) error {
	if len(data) == 0 {
		return nil
	}
	return e.ck(C.dsl_download(e.h, buffer, (*C.float)(unsafe.Pointer(&data[0])), C.size_t(len(data))))
}

// The passes of model/sph.SPH (model/sph/fluid.go:100-197), one C call each.
func (e *Engine) NN() error                    { return e.ck(C.dsl_build_neighbours(e.h)) }
func (e *Engine) DensityAll() error            { return e.ck(C.dsl_density_pass(e.h)) }
func (e *Engine) PressureAll() error           { return e.ck(C.dsl_pressure_pass(e.h)) }
func (e *Engine) ViscousAll() error            { return e.ck(C.dsl_viscous_pass(e.h)) }
func (e *Engine) GradientPressureForce() error { return e.ck(C.dsl_gradient_pressure_pass(e.h)) }
func (e *Engine) Update() error                { return e.ck(C.dsl_update_pass(e.h)) }
func (e *Engine) ExternalAll(f [3]float32) error {
	return e.ck(C.dsl_external_pass(e.h, (*C.float)(unsafe.Pointer(&f[0]))))
}
func (e *Engine) WCSPHStep(n int) error  { return e.ck(C.dsl_wcsph_step(e.h, C.int(n))) }
func (e *Engine) PCISPHBegin() error     { return e.ck(C.dsl_pcisph_begin(e.h)) }
func (e *Engine) PCISPHStep(n int) error { return e.ck(C.dsl_pcisph_step(e.h, C.int(n))) }
func (e *Engine) Sync() error            { return e.ck(C.dsl_sync(e.h)) }
func (e *Engine) SetParams() error       { return e.ck(C.dsl_set_params(e.h, &e.Params)) }

// ---- boundary particles (model/particle_array.go:123-128, model/field/sph_field.go:75-85) ----

// AddBoundaryParticles appends position-only particles behind the fluid (needs room in Params.capacity).
func (e *Engine) AddBoundaryParticles(positions []float32) error {
	if len(positions) == 0 {
		return nil
	}
	if err := e.ck(C.dsl_add_boundary_particles(e.h, (*C.float)(unsafe.Pointer(&positions[0])), C.size_t(len(positions)))); err != nil {
		return err
	}
	// the library's n_boundary has grown: refresh the copy that SetParams sends back and that sizes the buffers
	return e.ck(C.dsl_get_params(e.h, &e.Params))
}

// ---- the SPHField operators no solver calls (model/field/sph_field.go:124-135,203-294) ----

func (e *Engine) fieldOut(n int, call func(*C.float, C.size_t) C.int) ([]float32, error) {
	out := make([]float32, n)
	if n == 0 {
		return out, nil
	}
	return out, e.ck(call((*C.float)(unsafe.Pointer(&out[0])), C.size_t(n)))
}
func (e *Engine) Div(tensorBuffer C.int) ([]float32, error) {
	return e.fieldOut(int(e.Params.n_particles), func(p *C.float, n C.size_t) C.int { return C.dsl_field_divergence(e.h, tensorBuffer, p, n) })
}
func (e *Engine) Curl(tensorBuffer C.int) ([]float32, error) {
	return e.fieldOut(3*int(e.Params.n_particles), func(p *C.float, n C.size_t) C.int { return C.dsl_field_curl(e.h, tensorBuffer, p, n) })
}
func (e *Engine) Laplacian(scalarBuffer C.int) ([]float32, error) {
	return e.fieldOut(int(e.Params.n_particles), func(p *C.float, n C.size_t) C.int { return C.dsl_field_laplacian(e.h, scalarBuffer, p, n) })
}
func (e *Engine) Interpolate(scalarBuffer C.int, positions []float32) ([]float32, error) {
	out := make([]float32, len(positions)/3)
	if len(out) == 0 {
		return out, nil
	}
	return out, e.ck(C.dsl_field_interpolate(e.h, scalarBuffer, (*C.float)(unsafe.Pointer(&positions[0])), C.size_t(len(out)),
		(*C.float)(unsafe.Pointer(&out[0]))))
}

// ---- render hand-off without the per-step full read-back (pcisph_gpu_darwin.go:276-282) ----

// DownloadDecimated: every stride-th particle of positions / velocities.
func (e *Engine) DownloadDecimated(buffer C.int, stride int, out []float32) error {
	if len(out) == 0 {
		return nil
	}
	return e.ck(C.dsl_download_decimated(e.h, buffer, C.int(stride), (*C.float)(unsafe.Pointer(&out[0])), C.size_t(len(out))))
}

// DevicePointers: the live SoA device arrays (x, y, z), the slot -> particle id map and the slot count.
func (e *Engine) DevicePointers(buffer C.int) (xyz [3]unsafe.Pointer, ids unsafe.Pointer, n int, err error) {
	var p [3]*C.float
	var i *C.int32_t
	var cn C.int
	err = e.ck(C.dsl_device_pointers(e.h, buffer, (**C.float)(unsafe.Pointer(&p[0])), &i, &cn))
	return [3]unsafe.Pointer{unsafe.Pointer(p[0]), unsafe.Pointer(p[1]), unsafe.Pointer(p[2])}, unsafe.Pointer(i), int(cn), err
}

// ---- lsh_ref parity mode (sampler/lsh/lsh.go) ----

func (e *Engine) SetHashVectors(vectors []float32) error { // lsh.Allocate's random projections, bits x 3
	return e.ck(C.dsl_set_hash_vectors(e.h, (*C.float)(unsafe.Pointer(&vectors[0])), C.int(len(vectors)/3)))
}

// ---- multi-GPU: one Engine per device, RCCL inside the library (include/dslsph.h) ----

// Comm wraps dsl_comm.  Rank 0 calls CommUniqueID and hands the 128 bytes to the other ranks over any channel.
type Comm struct{ c *C.dsl_comm }

func CommUniqueID() ([128]byte, error) {
	var id [128]byte
	if rc := C.dsl_comm_unique_id((*C.uint8_t)(unsafe.Pointer(&id[0]))); rc != 0 {
		return id, errors.New(C.GoString(C.dsl_comm_last_error()))
	}
	return id, nil
}
func NewComm(nranks, rank int, id [128]byte, device int) (*Comm, error) {
	c := &Comm{}
	if rc := C.dsl_comm_create(C.int(nranks), C.int(rank), (*C.uint8_t)(unsafe.Pointer(&id[0])), C.int(device), &c.c); rc != 0 {
		return nil, errors.New(C.GoString(C.dsl_comm_last_error()))
	}
	return c, nil
}
func (c *Comm) Close() { C.dsl_comm_destroy(c.c); c.c = nil }

// Count: the ranks the communicator really spans (ncclCommCount).
func (c *Comm) Count() (int, error) {
	var n C.int
	if rc := C.dsl_comm_count(c.c, &n); rc != 0 {
		return 0, errors.New(C.GoString(C.dsl_comm_last_error()))
	}
	return int(n), nil
}

// Transport: the host's own transport behind NewCommCustom (dsl_transport in include/dslsph.h).  The callbacks are C
// function pointers -- exported Go functions via //export, or plain C -- held as unsafe.Pointer so that packages other
// than this one can fill the table (cgo's C.dsl_transport is private to the package that imports "C").
type Transport struct {
	Ctx             unsafe.Pointer
	GroupStart      unsafe.Pointer // int (*)(void *ctx)
	GroupEnd        unsafe.Pointer // int (*)(void *ctx)
	Send            unsafe.Pointer // int (*)(void *ctx, const void *dev_buf, size_t bytes, int peer, void *stream)
	Recv            unsafe.Pointer // int (*)(void *ctx, void *dev_buf, size_t bytes, int peer, void *stream)
	AllReduceMaxU32 unsafe.Pointer // int (*)(void *ctx, void *dev_buf, size_t count, void *stream)
}

// NewCommCustom: a communicator over the host's own transport (MPI, sockets, ...) instead of RCCL.  The library calls
// the table's callbacks in exactly the order it would call RCCL; the table is copied.
func NewCommCustom(nranks, rank, device int, t Transport) (*Comm, error) {
	var table C.dsl_transport
	table.ctx = t.Ctx
	*(*unsafe.Pointer)(unsafe.Pointer(&table.group_start)) = t.GroupStart
	*(*unsafe.Pointer)(unsafe.Pointer(&table.group_end)) = t.GroupEnd
	*(*unsafe.Pointer)(unsafe.Pointer(&table.send)) = t.Send
	*(*unsafe.Pointer)(unsafe.Pointer(&table.recv)) = t.Recv
	*(*unsafe.Pointer)(unsafe.Pointer(&table.all_reduce_max_u32)) = t.AllReduceMaxU32
	c := &Comm{}
	if rc := C.dsl_comm_create_custom(C.int(nranks), C.int(rank), C.int(device), &table, &c.c); rc != 0 {
		return nil, errors.New(C.GoString(C.dsl_comm_last_error()))
	}
	return c, nil
}

// NewEngines: one process, several devices (dsl_create_multi: handles + ncclCommInitAll).  Drive each Engine
// from its own goroutine under runtime.LockOSThread: RCCL wants one host thread per device.
func NewEngines(params []C.dsl_params, devices []int) ([]*Engine, []*Comm, error) {
	n := len(devices)
	devs := make([]C.int, n)
	for k, d := range devices {
		devs[k] = C.int(d)
	}
	hs := make([]*C.dsl_handle, n)
	cs := make([]*C.dsl_comm, n)
	if rc := C.dsl_create_multi(&params[0], C.int(n), &devs[0], &hs[0], &cs[0]); rc != 0 {
		return nil, nil, lastError(nil)
	}
	engs, comms := make([]*Engine, n), make([]*Comm, n)
	for k := range hs {
		engs[k], comms[k] = &Engine{h: hs[k], Params: params[k]}, &Comm{c: cs[k]}
	}
	return engs, comms, nil
}

// SlabConfig / SlabAttach: this engine owns [lo, hi) along axis and exchanges a 2h band with the ranks lo_rank
// and hi_rank (-1 at a domain end) every step.
func (e *Engine) SlabConfig(axis int, lo, hi float32) error {
	return e.ck(C.dsl_slab_config(e.h, C.int(axis), C.float(lo), C.float(hi)))
}
func (e *Engine) SetIDs(ids []int32) error {
	return e.ck(C.dsl_set_ids(e.h, (*C.int32_t)(unsafe.Pointer(&ids[0])), C.size_t(len(ids))))
}
func (e *Engine) SlabAttach(c *Comm, loRank, hiRank int, widthFull, width float32, capFull, capX int, overlap bool) error {
	var cc *C.dsl_comm
	if c != nil {
		cc = c.c
	}
	ov := C.int(0)
	if overlap {
		ov = 1
	}
	return e.ck(C.dsl_slab_attach(e.h, cc, C.int(loRank), C.int(hiRank), C.float(widthFull), C.float(width), C.int(capFull), C.int(capX), ov))
}
func (e *Engine) SlabWCSPHStep(n int) error  { return e.ck(C.dsl_slab_wcsph_step(e.h, C.int(n))) }
func (e *Engine) SlabPCISPHStep(n int) error { return e.ck(C.dsl_slab_pcisph_step(e.h, C.int(n))) }
func (e *Engine) SlabReplan() error          { return e.ck(C.dsl_slab_replan(e.h)) }

// MaxV: SPH.MaxV() (fluid.go:206)
func (e *Engine) MaxV() (float32, error) {
	var st C.dsl_stats
	if err := e.ck(C.dsl_get_stats(e.h, &st)); err != nil {
		return 0, err
	}
	return float32(st.max_vel), nil
}

// ---------------------------------------------------------------------------------------
// compute/gpu.ComputeGPU facade (compute/gpu/gpu.go:20-425, compute/compute.go:26-53)
// ---------------------------------------------------------------------------------------

// Descriptor mirrors compute.Descriptor (compute/compute.go:9-13).
type Descriptor struct {
	Work  []int
	Local []int
	Size  int
}

type ComputeGPU struct {
	desc       *Descriptor
	eng        *Engine
	registered map[string]int
	kernels    map[string]bool
	pending    string
	log        string
}

func New_ComputeGPU(desc *Descriptor, eng *Engine) *ComputeGPU {
	return &ComputeGPU{desc: desc, eng: eng, registered: map[string]int{}, kernels: map[string]bool{}}
}

func (cp *ComputeGPU) RegisterBuffer(bytes_size int, t int, name string) error { // gpu.go:314-321
	if auxBuffers[name] {
		return nil
	}
	if _, ok := bufferIDs[name]; !ok {
		return fmt.Errorf("buffer [%s] is not a buffer of this engine", name)
	}
	cp.registered[name] = bytes_size
	return nil
}

func (cp *ComputeGPU) isregistered(name string) error { // gpu.go:305-310
	if _, ok := cp.registered[name]; !ok {
		return fmt.Errorf("buffer [%s] not registered", name)
	}
	return nil
}

func (cp *ComputeGPU) PassFloatBuffer(cpu_buffer []float32, name string) error { // gpu.go:343-352
	if auxBuffers[name] {
		return nil
	}
	if err := cp.isregistered(name); err != nil {
		return err
	}
	return cp.eng.Upload(bufferIDs[name], cpu_buffer)
}

func (cp *ComputeGPU) ReadFloatBuffer(cpu_buffer []float32, name string) error { // gpu.go:332-341
	if err := cp.isregistered(name); err != nil {
		return err
	}
	return cp.eng.Download(bufferIDs[name], cpu_buffer)
}

// PassLayoutBuffer (gpu.go:378-390) carries the two parameter blocks of pcisph_gpu_darwin.go:60-61.  They
// are not dropped: "sizes" {N, Nboundary, buckets, bucket_size} and "floats" {dt, mass, delta, maxVel, h}
// are checked against the engine's own parameters, so a host that disagrees with the device finds out.
func (cp *ComputeGPU) PassLayoutBuffer(data interface{}, bytes int, name string) error {
	if !auxBuffers[name] {
		return fmt.Errorf("buffer [%s] not registered", name)
	}
	p := &cp.eng.Params
	switch v := data.(type) {
	case []int32:
		if name == "sizes" && len(v) >= 2 && (C.int32_t(v[0]) != p.n_particles || C.int32_t(v[1]) != p.n_boundary) {
			return errors.New("sizes block {N, Nboundary, ..} does not match the engine's parameters")
		}
	case []float32:
		if name == "floats" && len(v) >= 5 && (C.float(v[0]) != p.dt || C.float(v[1]) != p.mass || C.float(v[4]) != p.h) {
			return errors.New("floats block {dt, mass, delta, maxVel, h} does not match the engine's parameters")
		}
	}
	cp.log += "Passed Layout Buffer " + name + "\n"
	return nil
}
func (cp *ComputeGPU) AddSourceFile(filename string) error   { return nil }  // gpu.go:257-271: nothing to compile
func (cp *ComputeGPU) AddSourceString(source string) bool    { return true } // compute.go:46
func (cp *ComputeGPU) BuildProgram(include_dir string) error { return nil }  // gpu.go:194-229

// ---- the rest of compute.GPUCompute (compute/compute.go:26-53) ----

// Setup (compute.go:29): the device context is the handle NewEngine made.
func (cp *ComputeGPU) Setup(gl bool) bool { return cp.HasDeviceContext() }

// Queue (gpu.go:286-296) names the built-in kernel the next Run executes.
func (cp *ComputeGPU) Queue(name string) error {
	if !cp.kernels[name] {
		return fmt.Errorf("kernel [%s] not registered", name)
	}
	cp.pending = name
	return nil
}

// Run (compute.go:33): the reference's two fused device kernels are phases of the PCISPH step here --
// compute_density (pci_density.c:12-23) = NN, density, viscosity; predict_correct (pci_predict.c:9-27) = the
// correction loop + integrate -- then THREAD_DONE (103) or THREAD_ERR (102) goes down the channel.
func (cp *ComputeGPU) Run(x chan int) {
	var rc C.int
	h := cp.eng.h
	switch cp.pending {
	case "compute_density":
		if rc = C.dsl_pcisph_begin(h); rc == 0 {
			rc = C.dsl_pcisph_phase(h, C.DSL_PCI_BEGIN_STEP)
		}
	case "predict_correct":
		for it := 0; rc == 0 && it < int(cp.eng.Params.pci_max_iters); it++ {
			if rc = C.dsl_pcisph_phase(h, C.DSL_PCI_ITERATE); rc == 0 {
				rc = C.dsl_pcisph_phase(h, C.DSL_PCI_CHECK)
			}
		}
		if rc == 0 {
			rc = C.dsl_pcisph_phase(h, C.DSL_PCI_END_STEP)
		}
	default:
		rc = C.DSL_ERR_INVALID
	}
	if rc == 0 {
		rc = C.dsl_sync(h)
	}
	if rc == 0 {
		x <- 103
	} else {
		x <- 102
	}
}

// PassIntBuffer / ReadIntBuffer (gpu.go:354-378): "sizes" is checked / returned; "sampler" (the host's
// flattened LSH table, lsh.go:70-80) is accepted and ignored on the way in -- the engine builds its own
// neighbour table -- and is HashSampler.GetData1D of the device table on the way out (DSL_NEIGH_LSH_REF).
func (cp *ComputeGPU) PassIntBuffer(cpu_buffer []int, name string) error {
	if !auxBuffers[name] {
		return cp.isregistered(name)
	}
	p := &cp.eng.Params
	if name == "sizes" && len(cpu_buffer) >= 2 && (C.int32_t(cpu_buffer[0]) != p.n_particles || C.int32_t(cpu_buffer[1]) != p.n_boundary) {
		return errors.New("sizes block {N, Nboundary, ..} does not match the engine's parameters")
	}
	cp.log += "Passed Integer Buffer " + name + "\n"
	return nil
}
func (cp *ComputeGPU) ReadIntBuffer(cpu_buffer []int, name string) error {
	p := &cp.eng.Params
	switch name {
	case "sizes":
		v := [4]int{int(p.n_particles), int(p.n_boundary), int(p.lsh_buckets), int(p.lsh_bucket_size)}
		copy(cpu_buffer, v[:])
		return nil
	case "sampler":
		t := make([]int32, len(cpu_buffer))
		if len(t) == 0 {
			return nil
		}
		if err := cp.eng.ck(C.dsl_lsh_download_table(cp.eng.h, (*C.int32_t)(unsafe.Pointer(&t[0])), C.size_t(len(t)))); err != nil {
			return err
		}
		for k := range t {
			cpu_buffer[k] = int(t[k])
		}
		return nil
	}
	return fmt.Errorf("buffer [%s] holds no integers", name)
}

// RegisterGLBuffer (gpu.go:323-330) shares a GL buffer with the device queue so that the renderer draws what
// the solver wrote without a read-back.  The counterpart here goes the other way: the consumer is handed the
// live device arrays (DevicePointers); the GL id is only recorded.
func (cp *ComputeGPU) RegisterGLBuffer(gl_buffer_id uint32, size int, name string) error {
	if name != "positions" {
		return errors.New("only the positions buffer has a render hand-off")
	}
	cp.registered[name] = size
	cp.log += fmt.Sprintf("RegisterGLBuffer() - buffer %s (GL id %d): read the device arrays through DevicePointers()\n", name, gl_buffer_id)
	return nil
}

func (cp *ComputeGPU) RegisterKernel(name string) bool { // gpu.go:231-250
	ok := name == "compute_density" || name == "predict_correct"
	if ok {
		cp.kernels[name] = true
	}
	return ok
}
func (cp *ComputeGPU) Set(d Descriptor)       { *cp.desc = d }
func (cp *ComputeGPU) Get() Descriptor        { return *cp.desc }
func (cp *ComputeGPU) HasDeviceContext() bool { return cp.eng != nil && cp.eng.h != nil }
func (cp *ComputeGPU) ValidState() bool       { return cp.HasDeviceContext() }
func (cp *ComputeGPU) Log() string            { return cp.log }

// ---- the rest of include/dslsph.h: one method per export, no logic -------------------------------------

func Version() string { return C.GoString(C.dsl_version()) }

func (e *Engine) GetParams() (C.dsl_params, error) {
	var p C.dsl_params
	err := e.ck(C.dsl_get_params(e.h, &p))
	return p, err
}
// Library options (DSL_OPT_* in include/dslsph.h) a host may want to name: the neighbour-list skin of WCSPHStep and its
// read-only counters, how far ahead of the particles the lists are built, the tile kernels' grid oversubscription.
const (
	OptSkin            = int(C.DSL_OPT_SKIN)
	OptSkinSteps       = int(C.DSL_OPT_SKIN_STEPS)
	OptSkinRebuilds    = int(C.DSL_OPT_SKIN_REBUILDS)
	OptSkinSuspensions = int(C.DSL_OPT_SKIN_SUSPENSIONS)
	OptSkinPredict     = int(C.DSL_OPT_SKIN_PREDICT)
	OptDeviceBytes     = int(C.DSL_OPT_DEVICE_BYTES)
	OptGridOversub     = int(C.DSL_OPT_GRID_OVERSUB)
)

// SetOption / GetOption: library options (DSL_OPT_* in include/dslsph.h), e.g. the neighbour-list skin of WCSPHStep.
func (e *Engine) SetOption(option int, value float64) error {
	return e.ck(C.dsl_set_option(e.h, C.int(option), C.double(value)))
}
func (e *Engine) GetOption(option int) (float64, error) {
	var v C.double
	err := e.ck(C.dsl_get_option(e.h, C.int(option), &v))
	return float64(v), err
}

func (e *Engine) ResetForces() error { return e.ck(C.dsl_reset_forces(e.h)) } // the state Update leaves (fluid.go:193)
func (e *Engine) ForcePass() error   { return e.ck(C.dsl_force_pass(e.h)) }   // fused [G] [V] X U of the WCSPH step

// SetStream: order the engine's launches on the caller's HIP stream (nil = the null stream); UseOwnStream undoes it.
func (e *Engine) SetStream(hipStream unsafe.Pointer) error { return e.ck(C.dsl_set_stream(e.h, hipStream)) }
func (e *Engine) UseOwnStream() error                      { return e.ck(C.dsl_use_own_stream(e.h)) }

// Count: live particles (owned + ghosts) and owned particles of a slab engine; blocking.
func (e *Engine) Count() (live, owned int, err error) {
	var a, b C.int
	err = e.ck(C.dsl_get_count(e.h, &a, &b))
	return int(a), int(b), err
}

// Per-kernel timing (HIP events on the launch stream): mode 0 off, 1 every kernel, 2 the dominant kernels only.
func (e *Engine) TimingEnable(mode int) error { return e.ck(C.dsl_timing_enable(e.h, C.int(mode))) }
func (e *Engine) TimingReset() error          { return e.ck(C.dsl_timing_reset(e.h)) }
func (e *Engine) Timing(kernelID int) (avgMs float64, launches int64, err error) {
	var ms C.double
	var n C.int64_t
	err = e.ck(C.dsl_timing_get(e.h, C.int(kernelID), &ms, &n))
	return float64(ms), int64(n), err
}

// Slot-order views (tests, debugging): a 3-component buffer without the un-sort, the slot -> particle id map,
// the cell table of the current neighbour build.
func (e *Engine) DownloadSorted(buffer C.int, out []float32) error {
	return e.ck(C.dsl_download_sorted(e.h, buffer, (*C.float)(unsafe.Pointer(&out[0])), C.size_t(len(out))))
}
func (e *Engine) DownloadIDs(out []int32) error {
	return e.ck(C.dsl_download_ids(e.h, (*C.int32_t)(unsafe.Pointer(&out[0])), C.size_t(len(out))))
}
func (e *Engine) DownloadCellStart(out []int32) error {
	return e.ck(C.dsl_download_cell_start(e.h, (*C.int32_t)(unsafe.Pointer(&out[0])), C.size_t(len(out))))
}

// Slab pieces for a host that brings its own transport (INTEGRATION.md 7b); device pointers are the caller's.
func SlabMessageFloats(capFull, capX int) int {
	return int(C.dsl_slab_message_floats(C.int(capFull), C.int(capX)))
}
func (e *Engine) SlabMessageFloats(capFull, capX int) int {
	return int(C.dsl_slab_message_floats_for(e.h, C.int(capFull), C.int(capX)))
}
func (e *Engine) SlabRecordFloats() int { return int(C.dsl_slab_record_floats(e.h)) }
func (e *Engine) SlabSplit(width, margin float32) error {
	return e.ck(C.dsl_slab_split(e.h, C.float(width), C.float(margin)))
}
func (e *Engine) ForcePassSplit(phase int) error { return e.ck(C.dsl_force_pass_split(e.h, C.int(phase))) }
func (e *Engine) SlabPack(widthFull, width float32, devLo, devHi unsafe.Pointer, capFull, capX int) error {
	return e.ck(C.dsl_slab_pack(e.h, C.float(widthFull), C.float(width), (*C.float)(devLo), (*C.float)(devHi), C.int(capFull), C.int(capX)))
}
func (e *Engine) SlabPackBand(widthFull float32, devLo, devHi unsafe.Pointer, capFull, capX int, stream unsafe.Pointer) error {
	return e.ck(C.dsl_slab_pack_band(e.h, C.float(widthFull), (*C.float)(devLo), (*C.float)(devHi), C.int(capFull), C.int(capX), stream))
}
func (e *Engine) SlabAppend(devMsg unsafe.Pointer, capFull, capX int) error {
	return e.ck(C.dsl_slab_append(e.h, (*C.float)(devMsg), C.int(capFull), C.int(capX)))
}
func (e *Engine) SlabAppend2(devMsgA, devMsgB unsafe.Pointer, capFull, capX int) error {
	return e.ck(C.dsl_slab_append2(e.h, (*C.float)(devMsgA), (*C.float)(devMsgB), C.int(capFull), C.int(capX)))
}

// SlabStatus: [0] records that did not fit, [1] split margin outrun, [2], [3] band high-water marks.
func (e *Engine) SlabStatus(resetHighWater bool) (st [4]int32, err error) {
	r := C.int(0)
	if resetHighWater {
		r = 1
	}
	err = e.ck(C.dsl_slab_status(e.h, (*C.int32_t)(unsafe.Pointer(&st[0])), r))
	return st, err
}
func (e *Engine) SlabOverflow() (highWater int, err error) {
	var hw C.int
	err = e.ck(C.dsl_slab_overflow(e.h, &hw))
	return int(hw), err
}
func (e *Engine) SlabDetach() error   { return e.ck(C.dsl_slab_detach(e.h)) }
func (e *Engine) SlabExchange() error { return e.ck(C.dsl_slab_exchange(e.h)) }

// SlabImageShift: periodic images along the slab axis (a ring of ranks closes with -L / +L at its two ends).
func (e *Engine) SlabImageShift(fromLo, fromHi float32) error {
	return e.ck(C.dsl_slab_image_shift(e.h, C.float(fromLo), C.float(fromHi)))
}

// PCISPHErrorWord: copy the iteration's max density error out to / back in from a device word of the caller's
// (the all-reduce between DSL_PCI_ITERATE and DSL_PCI_CHECK of a host-driven slab step).
func (e *Engine) PCISPHErrorWord(devWord unsafe.Pointer, store bool) error {
	s := C.int(0)
	if store {
		s = 1
	}
	return e.ck(C.dsl_pcisph_error_word(e.h, (*C.uint32_t)(devWord), s))
}

// PCISPHSetBinning: -1 never, 0 automatic (default), 1 always -- whether the DensityF query points (the predictor's
// positions, which the reference never re-synchronises: pcisph_darwin.go:28-41) are sorted into grid cells of their own.
func (e *Engine) PCISPHSetBinning(mode int) error {
	return e.ck(C.dsl_pcisph_set_binning(e.h, C.int(mode)))
}

// PCISPHBinning: the mode, and whether the next correction iteration will sort its queries.
func (e *Engine) PCISPHBinning() (mode int, active bool, err error) {
	var m, a C.int
	err = e.ck(C.dsl_pcisph_get_binning(e.h, &m, &a, nil))
	return int(m), a != 0, err
}

// PCISPHQueryEscaped (slab mode, blocking): a query point of an owned particle has drifted more than h beyond a slab
// plane, out of what the 2h ghost band covers.
func (e *Engine) PCISPHQueryEscaped() (bool, error) {
	var x C.int
	err := e.ck(C.dsl_pcisph_get_binning(e.h, nil, nil, &x))
	return x != 0, err
}

// NewCommAll: one process, several devices (ncclCommInitAll); NewEngines wraps it together with the handles.
func NewCommAll(devices []int) ([]*Comm, error) {
	n := len(devices)
	devs := make([]C.int, n)
	for k, d := range devices {
		devs[k] = C.int(d)
	}
	cs := make([]*C.dsl_comm, n)
	if rc := C.dsl_comm_create_all(C.int(n), &devs[0], &cs[0]); rc != 0 {
		return nil, errors.New(C.GoString(C.dsl_comm_last_error()))
	}
	out := make([]*Comm, n)
	for k := range cs {
		out[k] = &Comm{c: cs[k]}
	}
	return out, nil
}
