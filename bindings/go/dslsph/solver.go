package dslsph

// Step drivers with the reference's signatures and channel protocol.

// Thread message enums, model/model.go:11-15.
const (
	THREAD_WAIT        = 100
	THREAD_GO          = 101
	THREAD_ERR         = 102
	THREAD_DONE        = 103
	SPH_THREAD_WAITING = 104
)

// WCSPH implements solver.SPHMethod (solver/method.go:3-6) like wcsph.WCSPH
// (solver/wcsph/wcsph.go:9-75); unlike the reference type it can be constructed.
type WCSPH struct{ core *Engine }

func NewWCSPH(core *Engine) WCSPH { return WCSPH{core} }

// Run: DensityAll, ExternalAll, PressureAll, Update, CFL forever (wcsph.go:14-26).
func (p WCSPH) Run() {
	for {
		if err := p.core.WCSPHStep(1); err != nil {
			return
		}
	}
}

// Run_: the same loop gated by the THREAD_* handshake (wcsph.go:35-75).
func (p WCSPH) Run_(t chan int) {
	sync := true
	for {
		if sync {
			if err := p.core.WCSPHStep(1); err != nil {
				t <- THREAD_ERR
				return
			}
			status := <-t
			if status == THREAD_WAIT {
				sync = false
				t <- SPH_THREAD_WAITING
				if waitStatus := <-t; waitStatus == THREAD_GO {
					sync = true
				}
			}
			if status == THREAD_GO {
				sync = true
			}
		}
		if waitStatus := <-t; waitStatus == THREAD_GO {
			sync = true
		}
	}
}

// PCISPH mirrors pcisph.PciMethod.Run (solver/pcisph/pcisph_darwin.go:24-118); the GL
// arguments of the reference signature belong to the renderer and are dropped.
type PCISPH struct{ system *Engine }

func NewPCIMethod(sys *Engine) *PCISPH { return &PCISPH{sys} }

func (pci *PCISPH) Run(message chan string) {
	if err := pci.system.PCISPHBegin(); err != nil { // :28-41
		return
	}
	done := false
	for !done {
		if err := pci.system.PCISPHStep(1); err != nil { // :43-101
			return
		}
		select { // :103-110
		case msg := <-message:
			if msg == "QUIT" {
				done = true
			}
		default:
		}
		select { // :112-116
		case message <- "SAMPLER_UPDATE":
		default:
		}
	}
}

// GPUPredictorCorrector mirrors solver/pcisph/pcisph_gpu_darwin.go:22-286.
type GPUPredictorCorrector struct {
	system      *Engine
	gpu_compute *ComputeGPU
	positions   []float32 // the host slice the renderer reads (ParticleArray.Positions())
}

func New_GPUPredictorCorrector(computeGPU *ComputeGPU, sys *Engine, positions []float32) (GPUPredictorCorrector, error) {
	m := GPUPredictorCorrector{system: sys, gpu_compute: computeGPU, positions: positions}
	n := int(sys.Params.n_particles)
	for name, bytes := range map[string]int{"positions": n * 12, "velocities": n * 12, "forces": n * 12,
		"densities": n * 4, "pressures": n * 4, "sizes": 16, "floats": 20, "temps": n * 28} { // :67-76
		if err := computeGPU.RegisterBuffer(bytes, 0, name); err != nil {
			return m, err
		}
	}
	if err := computeGPU.PassFloatBuffer(positions, "positions"); err != nil { // :81
		return m, err
	}
	if !computeGPU.RegisterKernel("compute_density") || !computeGPU.RegisterKernel("predict_correct") { // :133-139
		return m, errKernel
	}
	return m, sys.PCISPHBegin()
}

var errKernel = errorString("Register kernel failed")

type errorString string

func (e errorString) Error() string { return string(e) }

// Run: one cycle = compute_density + predict_correct (= one PCISPH step on the engine),
// blocking position read-back, "CL_REFRESH" (pcisph_gpu_darwin.go:249-286).
func (m GPUPredictorCorrector) Run(message *chan string) error {
	for {
		if err := m.system.PCISPHStep(1); err != nil { // :256-274
			return err
		}
		if err := m.gpu_compute.ReadFloatBuffer(m.positions, "positions"); err == nil { // :276-277
			*message <- "CL_REFRESH" // :279
		}
	}
}
