// What does each ingredient of fill_kernel cost on top of a plain copy?  (1e8 rows: 0.8 GB read + 0.8 GB written)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ bool bit_at(const uint32_t *bits, int64_t bit0, int64_t row) {
    const int64_t b = bit0 + row;
    return (bits[b >> 5] >> (b & 31)) & 1u;
}
// FEAT bit 0: per-row bit_at loads; bit 1: validity words out (shuffle + ballot + lane-0 stores); bit 2: shuffled neighbour values
// bit 3: validity by one 64-bit word load per lane-group instead of bit_at
template <int FEAT>
__global__ __launch_bounds__(256) void fill_v(const uint64_t *in, const uint32_t *vbits, uint64_t *out, uint32_t *out_words, int64_t n,
                                              unsigned long long *count) {
    unsigned long long nvalid = 0;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t base = wave * 512; base < n; base += nwaves * 512) {
        uint64_t a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int64_t g = base + 128 * k + 2 * lane;
            if (g + 1 < n) { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(in + g); a[k] = v.x; b[k] = v.y; } else { a[k] = 0; b[k] = 0; }
        }
        uint64_t mw[8];
        if (FEAT & 8) {
            // the trip's 512 validity bits = 8 aligned 64-bit words, the same for every lane (scalar-like loads)
            const uint64_t *w64 = reinterpret_cast<const uint64_t *>(vbits) + (base >> 6);
#pragma unroll
            for (int j = 0; j < 8; j++) mw[j] = w64[j];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int64_t cb = base + 128 * k, i = cb + 2 * lane;
            int va = 1, vb = 1;
            if (FEAT & 1) { va = (i < n && bit_at(vbits, 0, i)) ? 1 : 0; vb = (i + 1 < n && bit_at(vbits, 0, i + 1)) ? 1 : 0; }
            if (FEAT & 8) { const uint64_t w = mw[2 * k + (lane >> 5)]; va = (w >> ((2 * lane) & 63)) & 1; vb = (w >> ((2 * lane + 1) & 63)) & 1; }
            uint64_t oa = a[k], ob = b[k];
            if (FEAT & 4) {
                const int fin = va | (vb << 1);
                const int ilo = __shfl(fin, lane >> 1), ihi = __shfl(fin, 32 + (lane >> 1));
                const uint64_t m0 = __ballot((ilo >> (lane & 1)) & 1), m1 = __ballot((ihi >> (lane & 1)) & 1);
                for (int c = 0; c < 2; c++) {
                    const int r = 2 * lane + c;
                    uint64_t x = r >= 64 ? (m1 & ((1ull << (r - 64)) - 1ull)) : (m0 & ((1ull << r) - 1ull));
                    int qp = x ? (r >= 64 ? 127 : 63) - __clzll((long long)x) : (r >= 64 && m0 ? 63 - __clzll((long long)m0) : -1);
                    const int src = qp >= 0 ? (qp >> 1) : lane;
                    const uint64_t xx = __shfl((unsigned long long)a[k], src), yy = __shfl((unsigned long long)b[k], src);
                    const uint64_t pb = (qp & 1) ? yy : xx;
                    if (c == 0) { if (!va && qp >= 0) { oa = pb; va = 1; } } else { if (!vb && qp >= 0) { ob = pb; vb = 1; } }
                }
            }
            if (i + 1 < n) *reinterpret_cast<ulonglong2 *>(out + i) = make_ulonglong2(oa, ob);
            if (FEAT & 2) {
                const int f = va | (vb << 1);
                const int lo = __shfl(f, lane >> 1), hi = __shfl(f, 32 + (lane >> 1));
                const unsigned long long w0 = __ballot((lo >> (lane & 1)) & 1), w1 = __ballot((hi >> (lane & 1)) & 1);
                if (lane == 0 && cb < n) {
                    unsigned long long *dst = reinterpret_cast<unsigned long long *>(out_words + (cb >> 5));
                    dst[0] = w0;
                    if (cb + 64 < n) dst[1] = w1;
                    nvalid += __popcll(w0) + __popcll(w1);
                }
            }
        }
    }
    if (lane == 0 && nvalid) atomicAdd(count, nvalid);
}
template <int FEAT>
float run(const uint64_t *a, const uint32_t *vb, uint64_t *b, uint32_t *ow, int64_t n, unsigned long long *cnt, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(fill_v<FEAT>, dim3(grid), dim3(256), 0, 0, a, vb, b, ow, n, cnt);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
int main() {
    const int64_t n = 100000000ll;
    void *a, *b, *vb, *ow; unsigned long long *cnt;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&vb, n / 8 + 64)); CK(hipMalloc(&ow, n / 8 + 64)); CK(hipMalloc(&cnt, 8));
    CK(hipMemset(a, 1, n * 8)); CK(hipMemset(vb, 0xB7, n / 8 + 64));
    for (int grid : {4096, 16384}) {
        printf("grid %d\n", grid);
        printf("  copy only                         %.3f ms\n", run<0>((uint64_t *)a, (uint32_t *)vb, (uint64_t *)b, (uint32_t *)ow, n, cnt, grid));
        printf("  + bit_at per row                  %.3f ms\n", run<1>((uint64_t *)a, (uint32_t *)vb, (uint64_t *)b, (uint32_t *)ow, n, cnt, grid));
        printf("  + validity words out              %.3f ms\n", run<2>((uint64_t *)a, (uint32_t *)vb, (uint64_t *)b, (uint32_t *)ow, n, cnt, grid));
        printf("  + bit_at + words out              %.3f ms\n", run<3>((uint64_t *)a, (uint32_t *)vb, (uint64_t *)b, (uint32_t *)ow, n, cnt, grid));
        printf("  + bit_at + words + shuffles       %.3f ms\n", run<7>((uint64_t *)a, (uint32_t *)vb, (uint64_t *)b, (uint32_t *)ow, n, cnt, grid));
        printf("  word loads                        %.3f ms\n", run<8>((uint64_t *)a, (uint32_t *)vb, (uint64_t *)b, (uint32_t *)ow, n, cnt, grid));
        printf("  word loads + words out            %.3f ms\n", run<10>((uint64_t *)a, (uint32_t *)vb, (uint64_t *)b, (uint32_t *)ow, n, cnt, grid));
        printf("  word loads + words out + shuffles %.3f ms\n", run<14>((uint64_t *)a, (uint32_t *)vb, (uint64_t *)b, (uint32_t *)ow, n, cnt, grid));
    }
    return 0;
}
