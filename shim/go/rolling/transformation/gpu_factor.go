package transformation

import "reflect"

// transformation.Func is a plain func type (factor.go:5) and Factor(n) returns a closure over n (factor.go:7-20): nothing about a
// Func value says that it is a Factor, and a []Func (rolling/aggregation.go:46, :100-108) cannot hold a struct that would.  So the
// device path recognises a Factor by its CODE - every closure Factor returns shares one function literal, whose entry point
// reflect reports - and reads n back through the closure's own float64 branch: Factor(n)(1.0) is 1.0 * n, which is n bit for bit
// for every float64 (signed zeros, infinities and NaN payloads included).  No change to Factor itself.
var factorEntry = reflect.ValueOf(Factor(1)).Pointer()

// FactorOf reports the n of a Func built by Factor; ok is false for any other Func (the aggregator then stays on the Go path:
// rolling/gpu_cgo.go exportFactors).
func FactorOf(f Func) (n float64, ok bool) {
	if f == nil || reflect.ValueOf(f).Pointer() != factorEntry {
		return 0, false
	}
	r, err := f(float64(1))
	n, ok = r.(float64)
	return n, ok && err == nil
}
