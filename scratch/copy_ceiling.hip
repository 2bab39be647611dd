// Read / write / copy ceilings of the box (GB/s), 16-B accesses, to price the read+write kernels (fills, Interpolate) against.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void copy_k(const ulonglong2 *a, ulonglong2 *b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void copy4_k(const ulonglong2 *a, ulonglong2 *b, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    for (int64_t base = wave * 256; base < n; base += nw * 256) {
        ulonglong2 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) if (base + 64 * k + lane < n) v[k] = a[base + 64 * k + lane];
#pragma unroll
        for (int k = 0; k < 4; k++) if (base + 64 * k + lane < n) b[base + 64 * k + lane] = v[k];
    }
}
__global__ __launch_bounds__(256) void copynt_k(const ulonglong2 *a, ulonglong2 *b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        ulonglong2 v = a[i];
        __builtin_nontemporal_store(v.x, &b[i].x);
        __builtin_nontemporal_store(v.y, &b[i].y);
    }
}
__global__ __launch_bounds__(256) void write_k(ulonglong2 *b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) b[i] = make_ulonglong2(i, i);
}
__global__ __launch_bounds__(256) void read_k(const ulonglong2 *a, int64_t n, unsigned long long *out) {
    unsigned long long s = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { ulonglong2 v = a[i]; s += v.x ^ v.y; }
    if (s == 0x1234567) *out = s;
}
int main() {
    const int64_t bytes = 800000000ll, n = bytes / 16;
    void *a, *b; unsigned long long *o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 8));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {2048, 4096, 8192, 16384, 65536}) {
        for (int which = 0; which < 6; which++) {
            float best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0));
                if (which == 0) hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, (const ulonglong2 *)a, (ulonglong2 *)b, n);
                if (which == 1) hipLaunchKernelGGL(copy4_k, dim3(grid), dim3(256), 0, 0, (const ulonglong2 *)a, (ulonglong2 *)b, n);
                if (which == 2) hipLaunchKernelGGL(copynt_k, dim3(grid), dim3(256), 0, 0, (const ulonglong2 *)a, (ulonglong2 *)b, n);
                if (which == 3) hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, (ulonglong2 *)b, n);
                if (which == 4) hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, (const ulonglong2 *)a, n, o);
                if (which == 5) CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0));
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const char *nm[] = {"copy", "copy x4 in flight", "copy nontemporal store", "write only", "read only", "hipMemcpy D2D"};
            const double moved = (which == 3 || which == 4) ? bytes : 2.0 * bytes;
            printf("grid %6d  %-24s %.3f ms  %.0f GB/s moved\n", grid, nm[which], best, moved / best / 1e6);
        }
    }
    return 0;
}
