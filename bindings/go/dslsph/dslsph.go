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

func (cp *ComputeGPU) PassLayoutBuffer(data interface{}, bytes int, name string) error { return nil } // gpu.go:378-390
func (cp *ComputeGPU) AddSourceFile(filename string) error                           { return nil } // gpu.go:257-271
func (cp *ComputeGPU) BuildProgram(include_dir string) error                         { return nil } // gpu.go:194-229

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
