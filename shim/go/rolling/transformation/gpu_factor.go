package transformation

// Factor (factor.go:7-20) returns a closure; for the device path it returns a value that is still a Func AND exports its n, so that
// rolling/gpu_cgo.go can pass the chain to the library (bowgpu_agg.factors) instead of calling it once per window.
type factorFunc struct {
	Func
	n float64
}

func (f factorFunc) GPUFactor() (float64, bool) { return f.n, true }

// FactorGPU wraps what Factor(n) builds today: `return factorFunc{Func: <the existing closure>, n: n}` inside Factor itself.
func FactorGPU(existing Func, n float64) interface{ GPUFactor() (float64, bool) } {
	return factorFunc{existing, n}
}
