package rolling

import "github.com/metronlab/bow"

// GPU kind tags of the built-in aggregators: the values of include/bowgpu.h BOWGPU_AGG_* (checked by tests/test_go_shim.py).
const (
	GPUKindNone              int32 = -1 // a user closure: the reference's own Go path
	GPUKindWindowStart       int32 = 0
	GPUKindSum               int32 = 1
	GPUKindArithmeticMean    int32 = 2
	GPUKindMin               int32 = 3
	GPUKindMax               int32 = 4
	GPUKindCount             int32 = 5
	GPUKindFirst             int32 = 6
	GPUKindLast              int32 = 7
	GPUKindIntegralStep      int32 = 8
	GPUKindIntegralTrapezoid int32 = 9
	GPUKindWeightedAvgStep   int32 = 10
	GPUKindWeightedAvgLinear int32 = 11
	GPUKindMode              int32 = 13
)

// GPU kind tags of the built-in interpolators: include/bowgpu.h BOWGPU_INTERP_*.
const (
	GPUInterpWindowStart  int32 = 0
	GPUInterpLinear       int32 = 1
	GPUInterpStepPrevious int32 = 2
	GPUInterpNone         int32 = 3
)

// The tag lives IN the reference's own types (patches/0001: a `gpuKind int32` field in colAggregation, rolling/aggregation.go:41-51,
// and in the ColInterpolation struct, rolling/interpolation.go:10-16), stored as kind + 1 so that the zero value means "no tag".
// A field - not a wrapper type - because RenameOutput and SetTransformations return `aCopy := *a` (aggregation.go:82-86, :104-108):
// the copy keeps every field, so aggregation.ArithmeticMean("v").RenameOutput("m").SetTransformations(transformation.Factor(2))
// still carries GPUKindArithmeticMean when it reaches describe() in gpu_cgo.go.

// NewColAggregationGPU is NewColAggregation (rolling/aggregation.go:53-61) for the built-in constructors of rolling/aggregation;
// fn stays the reference's closure and is what runs whenever the device path declines.
func NewColAggregationGPU(inputName string, needInclusiveWindow bool, typ bow.Type, fn ColAggregationFunc, kind int32) ColAggregation {
	a := NewColAggregation(inputName, needInclusiveWindow, typ, fn).(*colAggregation)
	a.gpuKind = kind + 1
	return a
}

// gpuKindOfAggregation: the tag of a built-in aggregator, GPUKindNone for anything else (a user's own ColAggregation implementation,
// a closure handed to NewColAggregation).
func gpuKindOfAggregation(a ColAggregation) int32 {
	if ca, ok := a.(*colAggregation); ok {
		return ca.gpuKind - 1
	}
	return GPUKindNone
}

// NewColInterpolationGPU is NewColInterpolation (rolling/interpolation.go:22-28) plus the tag; ColInterpolation is a struct passed
// by value, so the field travels with every copy (Interpolate(interps ...ColInterpolation), validateInterpolation(&interps[i], i)).
func NewColInterpolationGPU(colName string, inputTypes []bow.Type, fn ColInterpolationFunc, kind int32) ColInterpolation {
	ip := NewColInterpolation(colName, inputTypes, fn)
	ip.gpuKind = kind + 1
	return ip
}
