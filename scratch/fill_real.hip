// The real fill_kernel (bow_amd/csrc/interp_fill.hip, included as source) timed on 1e8 rows with 30 % random nulls.
#include "../bow_amd/csrc/interp_fill.hip"
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <vector>
#include <random>
namespace bowgpu {
void set_error(const char *, ...) {}
int fail(int code, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return code; }
int hip_fail(hipError_t e, const char *what) { printf("%s: %s\n", what, hipGetErrorString(e)); return -100; }
int ctx_pool(Ctx *, int, size_t bytes, void **dptr) { return hipMalloc(dptr, bytes) == hipSuccess ? 0 : -100; }  // (leaks: a harness)
}
using namespace bowgpu;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int64_t n = 100000000ll;
    void *a, *ref, *b, *vb, *ow, *work; unsigned long long *cnt;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&ref, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&vb, n / 8 + 64)); CK(hipMalloc(&ow, n / 8 + 64)); CK(hipMalloc(&cnt, 8));
    CK(hipMemset(a, 1, n * 8)); CK(hipMemset(ref, 1, n * 8));
    std::vector<uint8_t> hv(n / 8 + 64);
    std::mt19937_64 rng(7);
    for (auto &x : hv) { uint8_t v = 0; for (int j = 0; j < 8; j++) v |= ((rng() % 10) >= 3 ? 1 : 0) << j; x = v; }
    CK(hipMemcpy(vb, hv.data(), hv.size(), hipMemcpyHostToDevice));
    Ctx c; c.stream = 0;
    CK(hipMalloc(&work, nbr_index_bytes(n, 0)));
    FillParams P; memset(&P, 0, sizeof P);
    if (nbr_index_build(&c, (const uint32_t *)vb, 0, n, work, &P.nbr)) return 1;
    P.ref_values = (const uint64_t *)ref; P.ref_vbits = nullptr; P.ref_type = BOWGPU_INT64;
    P.fill_values = (const uint64_t *)a; P.fill_vbits = (const uint32_t *)vb; P.fill_vbit0 = 0; P.fill_type = BOWGPU_FLOAT64;
    P.n = n; P.out_values = (uint64_t *)b; P.out_valid_words = (uint32_t *)ow; P.valid_count = cnt;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"FillPrevious", "FillNext", "FillMean", "FillLinear"};
    const int methods[] = {BOWGPU_FILL_PREVIOUS, BOWGPU_FILL_NEXT, BOWGPU_FILL_MEAN, kFillLinear};
    for (int pat = 0; pat < 2; pat++) {
    if (pat == 1) { CK(hipMemset(vb, 0xB7, n / 8 + 64)); if (nbr_index_build(&c, (const uint32_t *)vb, 0, n, work, &P.nbr)) return 1; printf("periodic nulls (every neighbour inside the chunk)\n"); }
    for (int m = 0; m < 4; m++) {
        P.method = methods[m];
        for (int grid : {2048}) {
            float best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0));
                (void)grid; if (fill_run(&c, P)) return 1;
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("%-14s grid %5d  %.3f ms\n", names[m], grid, best);
        }
    }
    }
    return 0;
}
