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

// gpuKinded is implemented by the aggregators / interpolators the built-in constructors return.
type gpuKinded interface{ GPUKind() int32 }

type kindedAggregation struct {
	ColAggregation
	kind int32
}

func (k kindedAggregation) GPUKind() int32 { return k.kind }

// NewColAggregationGPU is NewColAggregation (rolling/aggregation.go:53) plus the tag; fn stays the reference's closure and is
// what runs whenever the device path declines.
func NewColAggregationGPU(inputName string, needInclusiveWindow bool, typ bow.Type, fn ColAggregationFunc, kind int32) ColAggregation {
	return kindedAggregation{NewColAggregation(inputName, needInclusiveWindow, typ, fn), kind}
}

type kindedInterpolation struct {
	ColInterpolation
	kind int32
}

func (k kindedInterpolation) GPUKind() int32 { return k.kind }

// NewColInterpolationGPU is NewColInterpolation (rolling/interpolation.go:22-28) plus the tag.
func NewColInterpolationGPU(inputName string, inputTypes []bow.Type, fn ColInterpolationFunc, kind int32) ColInterpolation {
	return kindedInterpolation{NewColInterpolation(inputName, inputTypes, fn), kind}
}
