// headline_ab.hip — design-space probe for the benched shape (IntervalRolling(10) + WindowStart + ArithmeticMean, dense
// Float64, no nulls): variants of the wave-tile kernel and ablations of it, interleaved rounds in ONE process
// (cdna_hip_programming.md §5.4 rule 24), every variant checked against variant 0's output checksum.
// Not product code: the winning structure is ported into bow_amd/csrc/rolling_simple.hip.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scratch/headline_ab.hip -o scratch/bin/headline_ab
//   scratch/bin/headline_ab [rows=1e9] [rounds=7] [only=<variant substring>]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ulonglong2 nt_load(const ulonglong2 *q) {
    const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(q));
    return make_ulonglong2(v.x, v.y);
}

struct P {
    const int64_t *ts;
    const uint64_t *val;
    int64_t n, s0, interval, W;
    uint32_t m32, sh1, sh2;
    uint64_t *out_ws, *out_mean;
    uint32_t *status;
};

__device__ __forceinline__ uint32_t mdiv32(uint32_t n, uint32_t m, uint32_t sh1, uint32_t sh2) {
    const uint32_t t = __umulhi(m, n);
    return (t + ((n - t) >> sh1)) >> sh2;
}
__device__ __forceinline__ uint32_t left32(uint32_t x, uint32_t lane0) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ void lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------------------------
// TILE rows owned + 128 look-ahead rows per wavefront.  VDMA: the value column goes HBM -> LDS by global_load_lds_dwordx4
// (never through VGPRs).  ABL: 1 loads only, 2 + ids / heads / segment list, 3 + walk, 4 + stores (the full kernel).
// OST: results staged through LDS and stored 16 B per lane.  NT: non-temporal (aux = 2 / __builtin_nontemporal) loads.
template <int TILE, int VDMA, int ABL, int OST, int NT, int WPB, int ALIGN = 0, int NTS = 0>
__global__ __launch_bounds__(64 * WPB, TILE <= 640 ? (TILE == 512 ? 6 : (TILE < 512 ? 7 : 5)) / (WPB > 4 ? 2 : 1) : 3) void tile_kernel(const P p, const int64_t ntiles, const int64_t tiles_per_xcd) {
    constexpr int HALO = 128, ROWS = TILE + HALO, CH = ROWS / 128, SEGCAP = TILE <= 640 ? (TILE < 512 ? 300 : 400) : 800;
    struct Sh {
        uint64_t val[ROWS];
        uint32_t seg[SEGCAP + 2];
        uint64_t ost[OST ? 2 * (TILE / 4) : 1];
    };
    __shared__ Sh shs[WPB];
    Sh &sh = shs[threadIdx.x >> 6];
    const int64_t b = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int64_t tile = (b & 7) * tiles_per_xcd + (b >> 3);
    if (tile >= ntiles || (b >> 3) >= tiles_per_xcd) return;
    const int lane = threadIdx.x & 63;
    const int64_t base = tile * TILE;
    const int64_t n = p.n;
    const bool interior = base + ROWS <= n;
    const int nloc = interior ? ROWS : (int)(n - base);

    uint64_t ta[CH], tb[CH], va[CH], vb[CH];
    const uint64_t *__restrict__ tsq = reinterpret_cast<const uint64_t *>(p.ts);
    auto load_regs = [&](const uint64_t *__restrict__ src, uint64_t (&a)[CH], uint64_t (&bb)[CH]) {
#pragma unroll
        for (int j = 0; j < CH; j++) {
            int64_t r = base + j * 128 + 2 * lane;
            if (!interior && r + 2 > n) r = n - 2 > 0 ? (n - 2) & ~1ll : 0;   // in-bounds dummy (rows >= nloc are never used)
            const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(src + r);
            ulonglong2 x;
            if (NT == 1 || (NT == 2 && j > 0 && j < CH - 1)) x = nt_load(q); else x = *q;
            a[j] = x.x; bb[j] = x.y;
        }
    };
    load_regs(tsq, ta, tb);
    if (VDMA) {
#pragma unroll
        for (int j = 0; j < CH; j++) {
            int64_t r = base + j * 128 + 2 * lane;
            if (!interior && r + 2 > n) r = n - 2 > 0 ? (n - 2) & ~1ll : 0;
            if (NT == 1 || (NT == 2 && j > 0 && j < CH - 1))
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p.val + r),
                                                 (__attribute__((address_space(3))) void *)(&sh.val[j * 128]), 16, 0, 2);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p.val + r),
                                                 (__attribute__((address_space(3))) void *)(&sh.val[j * 128]), 16, 0, 0);
        }
    } else {
        load_regs(p.val, va, vb);
    }
    if (ABL == 1) {
        uint64_t x = 0;
#pragma unroll
        for (int j = 0; j < CH; j++) x ^= ta[j] ^ tb[j] ^ (VDMA ? 0 : va[j] ^ vb[j]);
        if (VDMA) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); x ^= sh.val[lane]; }
        if (x == 0x0123456789abcdefull) atomicOr(&p.status[7], 1u);
        return;
    }
    const int64_t left0 = base > 0 ? p.ts[base - 1] : INT64_MIN;
    const uint32_t s0_lo = (uint32_t)p.s0;
    bool unsorted = false, sat = false;
    const uint32_t w_first = mdiv32((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ta[0]) - s0_lo, p.m32, p.sh1, p.sh2);
    uint32_t left_w = base > 0 ? mdiv32((uint32_t)left0 - s0_lo, p.m32, p.sh1, p.sh2) : 0xFFFFFFFEu;
    int64_t left_ts = left0;
    int nseg_total = 0, nseg_owned = 0;
#pragma unroll
    for (int j = 0; j < CH; j++) {
        const int l = j * 128 + 2 * lane;
        const bool pa = l < nloc, pb = l + 1 < nloc;
        const int64_t tsa = (int64_t)ta[j], tsb = (int64_t)tb[j];
        const uint32_t plo = left32((uint32_t)tb[j], (uint32_t)left_ts);
        const uint32_t phi = left32((uint32_t)(tb[j] >> 32), (uint32_t)((uint64_t)left_ts >> 32));
        const int64_t prev_ts = (int64_t)(((uint64_t)phi << 32) | plo);
        unsorted |= (pa && prev_ts > tsa) || (pb && tsa > tsb);
        const uint32_t wa = mdiv32((uint32_t)tsa - s0_lo, p.m32, p.sh1, p.sh2);
        const uint32_t wb = mdiv32((uint32_t)tsb - s0_lo, p.m32, p.sh1, p.sh2);
        const uint32_t wprev = left32(wb, left_w);
        const bool ha = pa && (wa != wprev);
        const bool hb = pb && (wb != wa);
        const uint32_t la = wa - w_first, lb = wb - w_first;
        sat |= (ha && la >= 0xFFFFu) || (hb && lb >= 0xFFFFu);
        const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
        int pos = nseg_total;
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0));
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0));
        if (ha && pos < SEGCAP) sh.seg[pos] = (uint32_t)l | (la << 16);
        pos += ha ? 1 : 0;
        if (hb && pos < SEGCAP) sh.seg[pos] = (uint32_t)(l + 1) | (lb << 16);
        nseg_total += __popcll(ma) + __popcll(mb);
        if (j == CH - 2) nseg_owned = nseg_total;
        left_w = (uint32_t)__builtin_amdgcn_readlane((int)wb, 63);
        left_ts = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(tb[j] >> 32), 63) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tb[j], 63));
        if (!VDMA) *reinterpret_cast<ulonglong2 *>(&sh.val[l]) = make_ulonglong2(va[j], vb[j]);
    }
    if (NTS == 3) {   // diagnostic: the tile's stores issued EARLY (right after the loads returned), dense-shape slots, dummy values
        const int64_t slot0 = (base + 9) / 10, slot1 = (base + TILE + 9) / 10;
        for (int64_t sl = slot0 + lane; sl < slot1; sl += 64) { p.out_ws[sl] = ta[0]; p.out_mean[sl] = tb[0]; }
    }
    if (__ballot(unsorted)) { if (lane == 0) atomicOr(&p.status[0], 1u); return; }
    if (nseg_total > SEGCAP) sat = true;
    if (__ballot(sat)) { if (lane == 0) atomicOr(&p.status[4], 1u); return; }
    if (VDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_order();
    if (ABL == 2) {
        uint64_t x = sh.val[lane] ^ sh.seg[lane & 31];
        if (x == 0x0123456789abcdefull) atomicOr(&p.status[7], 1u);
        return;
    }
    const bool reaches_end = base + ROWS >= n;
    const uint32_t W32 = p.W > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)p.W;
    const int slot_first = (int)0;
    (void)slot_first;
    // ALIGN: a wavefront owns the windows whose slots lie in [ceilA(first head of its tile), ceilA(first head of the next tile)),
    // so no two wavefronts ever store into the same ALIGN * 8 bytes.  Both neighbours decide from the same 128 rows (the
    // left one's look-ahead = the right one's first chunk) whether the hand-over happens: the head that ends the handed-over
    // windows must lie inside those rows.
    int q_start = 0, q_end = nseg_owned;
    if (ALIGN) {
        auto handover = [&](int qf, int qlim) -> int {   // heads qf.. (qf = first head of a tile): index of the first one with an aligned id, or qf
            if (qf >= qlim) return qf;
            const uint32_t wf = w_first + (sh.seg[qf] >> 16);
            const uint32_t A = (wf + (ALIGN - 1)) & ~(uint32_t)(ALIGN - 1);
            if (A == wf) return qf;
            // heads are in id order: count those below A among qf .. qf + ALIGN (ids grow by at least one per head)
            const int qi = qf + lane;
            const bool below = qi < qlim && lane < ALIGN && (w_first + (sh.seg[qi < qlim ? qi : qf] >> 16)) < A;
            const int nb = __popcll(__ballot(below));
            return qf + nb < qlim ? qf + nb : qf;    // the closing head is not among the loaded rows: no hand-over
        };
        // my first chunk (rows < 128): heads of it = those with row < 128
        int n128 = 0;
        {
            const int qa = lane, qb = lane + 64;
            const bool a = qa < nseg_total && (int)(sh.seg[qa < nseg_total ? qa : 0] & 0xFFFFu) < 128;
            const bool bq = qb < nseg_total && (int)(sh.seg[qb < nseg_total ? qb : 0] & 0xFFFFu) < 128;
            n128 = __popcll(__ballot(a)) + __popcll(__ballot(bq));
        }
        if (tile > 0) q_start = handover(0, n128);
        q_end = handover(nseg_owned, nseg_total);
    }
    for (int q0 = q_start; q0 < q_end; q0 += 64) {
        const int q = q0 + lane;
        const bool act = q < q_end;
        uint64_t mean_bits = 0, ws_bits = 0;
        uint32_t wid = 0;
        bool store = false;
        if (act) {
            const uint32_t e0 = sh.seg[q], e1 = sh.seg[q + 1];
            const int r0 = (int)(e0 & 0xFFFFu);
            wid = w_first + (e0 >> 16);
            int r1 = -1;
            if (q + 1 < nseg_total) r1 = (int)(e1 & 0xFFFFu);
            else if (reaches_end) r1 = nloc;
            if (r1 >= 0 && wid < W32) {
                double sum = 0.0;
                int r = r0;
                for (; r + 4 <= r1; r += 4) {
                    const uint64_t x0 = sh.val[r], x1 = sh.val[r + 1], x2 = sh.val[r + 2], x3 = sh.val[r + 3];
                    sum += __longlong_as_double((long long)x0); sum += __longlong_as_double((long long)x1);
                    sum += __longlong_as_double((long long)x2); sum += __longlong_as_double((long long)x3);
                }
                for (; r < r1; r++) sum += __longlong_as_double((long long)sh.val[r]);
                mean_bits = (uint64_t)__double_as_longlong(sum / (double)(int64_t)(r1 - r0));
                ws_bits = (uint64_t)(p.s0 + (int64_t)((uint64_t)wid * (uint64_t)(uint32_t)p.interval));
                store = true;
            }
        }
        if (ABL == 3) {
            if ((mean_bits ^ ws_bits) == 0x0123456789abcdefull) atomicOr(&p.status[7], 1u);
            continue;
        }
        if (!OST) {
            if (store) {
                if (NTS == 4) {
                    __hip_atomic_store(&p.out_ws[wid], ws_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&p.out_mean[wid], mean_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (NTS == 5) { p.out_mean[wid] = mean_bits; if (ws_bits == 0x0123456789abcdefull) p.out_ws[wid] = ws_bits; }
                else if (NTS == 3) { if ((mean_bits ^ ws_bits) == 0x0123456789abcdefull) atomicOr(&p.status[7], 1u); }
                else if (NTS == 2) { p.out_ws[wid & 255u] = ws_bits; p.out_mean[wid & 255u] = mean_bits; }   // diagnostic: no HBM write traffic
                else if (NTS) { __builtin_nontemporal_store(ws_bits, &p.out_ws[wid]); __builtin_nontemporal_store(mean_bits, &p.out_mean[wid]); }
                else { p.out_ws[wid] = ws_bits; p.out_mean[wid] = mean_bits; }
            }
        } else {
            // slots of this trip are contiguous from wid(q0) when the data has no empty windows (dense case); otherwise
            // direct stores.  Stage and store pairs of slots as 16 B per lane, aligned on even slots.
            const uint32_t wq0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)wid);
            const bool dense = __ballot(store && wid != wq0 + (uint32_t)lane) == 0ull;
            const unsigned long long sm = __ballot(store);
            const int cnt = __popcll(sm);
            if (!dense || sm != (cnt == 64 ? ~0ull : ((1ull << cnt) - 1))) {
                if (store) { p.out_ws[wid] = ws_bits; p.out_mean[wid] = mean_bits; }
            } else {
                lds_order();
                sh.ost[lane] = ws_bits;
                sh.ost[64 + lane] = mean_bits;
                lds_order();
                const int odd = (int)(wq0 & 1u);           // first slot is odd: lane 0 of the pair pass starts one slot later
                // single slots at the unaligned ends
                if (odd && lane == 0) { p.out_ws[wq0] = sh.ost[0]; p.out_mean[wq0] = sh.ost[64]; }
                const int npair = (cnt - odd) >> 1;
                if (lane < npair) {
                    const int i = odd + 2 * lane;
                    const ulonglong2 a = make_ulonglong2(sh.ost[i], sh.ost[i + 1]);
                    const ulonglong2 m = make_ulonglong2(sh.ost[64 + i], sh.ost[64 + i + 1]);
                    *reinterpret_cast<ulonglong2 *>(&p.out_ws[wq0 + i]) = a;
                    *reinterpret_cast<ulonglong2 *>(&p.out_mean[wq0 + i]) = m;
                }
                if (((cnt - odd) & 1) && lane == 0) {
                    const int i = cnt - 1;
                    p.out_ws[wq0 + i] = sh.ost[i]; p.out_mean[wq0 + i] = sh.ost[64 + i];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the achievable line for THIS traffic mix: read 16 B per row from two streams, write 1.6 B per row to two streams,
// trivial arithmetic (one xor per loaded word).  MODE 0: registers, one-shot tiles of 512 rows per wave; MODE 1: LDS-DMA.
// ST: 0 the dense shape's pattern (51 lanes x 8 B, unaligned ends); 1 same bytes as whole aligned 1 KB blocks (two of five tiles per
// stream); 2 slots rounded to 16-slot (128 B) boundaries, 8 B per lane; 3 the same with 16 B per lane; 4 no stores; 5 pattern 0 non-temporal
template <int MODE, int NT, int WPB, int ST = 0>
__global__ __launch_bounds__(64 * WPB) void rw_ceiling_kernel(const uint64_t *__restrict__ a, const uint64_t *__restrict__ b, uint64_t *__restrict__ o0,
                                                              uint64_t *__restrict__ o1, const int64_t ntiles, const int64_t tiles_per_xcd) {
    __shared__ uint64_t shs[WPB][MODE ? 1024 : 1];
    __shared__ uint64_t pad[(ST == 7 || ST == 8 || ST == 9) ? 832 : 1];   // 6.6 KB per wave: the tile kernel's occupancy (24 waves per CU)
    const int64_t w = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int64_t tile = (w & 7) * tiles_per_xcd + (w >> 3);
    if (tile >= ntiles || (w >> 3) >= tiles_per_xcd) return;
    const int lane = threadIdx.x & 63;
    const int64_t base = tile * 512;
    uint64_t x = 0, y = 0;
    if (MODE == 0) {
        const ulonglong2 *pa = reinterpret_cast<const ulonglong2 *>(a + base) + lane, *pb = reinterpret_cast<const ulonglong2 *>(b + base) + lane;
        ulonglong2 a0, a1, a2, a3, b0, b1, b2, b3;
        if (NT) {
            a0 = nt_load(pa); a1 = nt_load(pa + 64); a2 = nt_load(pa + 128); a3 = nt_load(pa + 192);
            b0 = nt_load(pb); b1 = nt_load(pb + 64); b2 = nt_load(pb + 128); b3 = nt_load(pb + 192);
        } else {
            a0 = pa[0]; a1 = pa[64]; a2 = pa[128]; a3 = pa[192];
            b0 = pb[0]; b1 = pb[64]; b2 = pb[128]; b3 = pb[192];
        }
        x = a0.x ^ a0.y ^ a1.x ^ a1.y ^ a2.x ^ a2.y ^ a3.x ^ a3.y;
        y = b0.x ^ b0.y ^ b1.x ^ b1.y ^ b2.x ^ b2.y ^ b3.x ^ b3.y;
    } else {
        uint64_t *sh = shs[threadIdx.x >> 6];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a + base + j * 128 + 2 * lane),
                                             (__attribute__((address_space(3))) void *)(&sh[j * 128]), 16, 0, NT ? 2 : 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(b + base + j * 128 + 2 * lane),
                                             (__attribute__((address_space(3))) void *)(&sh[512 + j * 128]), 16, 0, NT ? 2 : 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; j++) { x ^= sh[j * 64 + lane]; y ^= sh[512 + j * 64 + lane]; }
    }
    asm volatile("" :: "v"(x), "v"(y));   // every lane's loads stay (a lane that does not store would otherwise lose them)
    // 512 rows -> 51.2 slots of each output: lanes 0..50 store (the dense shape's store pattern, 8 B per lane)
    int64_t slot0 = (base + 9) / 10;
    int64_t slot1 = (base + 512 + 9) / 10;
    if (ST == 7 || ST == 8 || ST == 9) { pad[lane] = x; asm volatile("" ::: "memory"); y ^= pad[(lane + 1) & 63] & 1; }
    if (ST == 6 || ST == 8) {   // ~2500 cycles between the loads' return and the stores (the tile kernel's reduce phase)
        asm volatile("" :: "v"(x), "v"(y));
#pragma unroll
        for (int k = 0; k < 5; k++) __builtin_amdgcn_s_sleep(127);
    }
    if (ST == 0 || (ST >= 6 && ST != 9)) { if (slot0 + lane < slot1) { o0[slot0 + lane] = x; o1[slot0 + lane] = y; } }
    else if (ST == 5) { if (slot0 + lane < slot1) { __builtin_nontemporal_store(x, &o0[slot0 + lane]); __builtin_nontemporal_store(y, &o1[slot0 + lane]); } }
    else if (ST == 1) {
        const int64_t g = tile / 5, k = tile % 5;
        if (k < 4) {
            uint64_t *o = (k & 1) ? o1 : o0;
            *reinterpret_cast<ulonglong2 *>(&o[(g * 2 + (k >> 1)) * 128 + 2 * lane]) = make_ulonglong2(x, y);
        }
    } else if (ST == 2 || ST == 3 || ST == 9) {
        slot0 = (slot0 + 15) & ~15ll; slot1 = (slot1 + 15) & ~15ll;
        if (ST == 2 || ST == 9) { if (slot0 + lane < slot1) { o0[slot0 + lane] = x; o1[slot0 + lane] = y; } }
        else if (slot0 + 2 * lane < slot1) {
            *reinterpret_cast<ulonglong2 *>(&o0[slot0 + 2 * lane]) = make_ulonglong2(x, y);
            *reinterpret_cast<ulonglong2 *>(&o1[slot0 + 2 * lane]) = make_ulonglong2(y, x);
        }
    } else if (x + y == 0x0123456789abcdefull) o0[0] = x;
}

// ---------------------------------------------------------------------------------------------------------------
// Streaming wavefront: one wavefront owns a contiguous range of KT x 512 rows and walks it chunk by chunk, the next
// chunk's loads in flight (second register set) while the current one is reduced - so stores and their acknowledgements
// overlap the next loads, and nothing is read twice: a window that continues into the next chunk is CARRIED (its running
// sum / count go on in row order in the next iteration: bit-exact however long the window is).  The window open at the end
// of the range is finished by its owner reading on in 128-row steps.
// HOT: stores go to a 4 KB region (no HBM write traffic; diagnostic).  OSTG: results staged in LDS, flushed as aligned 16-B stores.
template <int HOT, int OSTG, int NTL, int DEFER = 0, int STG = 256>     // STG: staged results per output (power of two ring); flushed in bursts of STG / 2
__global__ __launch_bounds__(64) void stream_kernel(const P p, const int64_t nranges, const int64_t ranges_per_xcd, const int KT) {
    constexpr int TILE = 512, CH = 4;
    constexpr int SEGCAP = 512;
    constexpr int FL = STG / 2;
    __shared__ uint64_t s_val[TILE];
    __shared__ uint32_t s_seg[SEGCAP + 2];
    __shared__ uint64_t s_out[OSTG ? 2 * STG : 2];
    const int64_t b = blockIdx.x;
    const int64_t rg = (b & 7) * ranges_per_xcd + (b >> 3);
    if (rg >= nranges) return;
    const int lane = threadIdx.x;
    const int64_t n = p.n;
    const int64_t rbase = rg * (int64_t)KT * TILE;
    const int64_t rend = rbase + (int64_t)KT * TILE < n ? rbase + (int64_t)KT * TILE : n;
    const int nit = (int)((rend - rbase + TILE - 1) / TILE);
    const uint64_t *__restrict__ tsq = reinterpret_cast<const uint64_t *>(p.ts);
    const uint32_t s0_lo = (uint32_t)p.s0;
    const uint32_t W32 = p.W > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)p.W;

    uint64_t ta0[CH], tb0[CH], va0[CH], vb0[CH], ta1[CH], tb1[CH], va1[CH], vb1[CH];
    auto load_set = [&](int it, uint64_t (&ta)[CH], uint64_t (&tb)[CH], uint64_t (&va)[CH], uint64_t (&vb)[CH]) {
        const int64_t base = rbase + (int64_t)it * TILE;
        const bool full = base + TILE <= n;
#pragma unroll
        for (int j = 0; j < CH; j++) {
            int64_t r = base + j * 128 + 2 * lane;
            if (!full && r + 2 > n) r = n - 2 > 0 ? (n - 2) & ~1ll : 0;
            const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(tsq + r);
            const ulonglong2 x = NTL ? nt_load(q) : *q;
            ta[j] = x.x; tb[j] = x.y;
        }
#pragma unroll
        for (int j = 0; j < CH; j++) {
            int64_t r = base + j * 128 + 2 * lane;
            if (!full && r + 2 > n) r = n - 2 > 0 ? (n - 2) & ~1ll : 0;
            const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(p.val + r);
            const ulonglong2 x = NTL ? nt_load(q) : *q;
            va[j] = x.x; vb[j] = x.y;
        }
    };

    // loop-carried, wave-uniform
    const int64_t left0 = rbase > 0 ? p.ts[rbase - 1] : INT64_MIN;
    uint32_t left_w = rbase > 0 ? mdiv32((uint32_t)left0 - s0_lo, p.m32, p.sh1, p.sh2) : 0xFFFFFFFEu;
    int64_t left_ts = left0;
    double c_sum = 0.0;
    int c_cnt = 0;
    uint32_t c_wid = 0;
    bool c_open = false;
    bool bad = false;
    int stg_n = 0;             // staged results not yet flushed
    uint32_t stg_w0 = 0;       // slot of the first staged result

    bool pend = false;
    uint32_t pend_wid = 0;
    uint64_t pend_ws = 0, pend_mean = 0;
    auto store_pair = [&](uint32_t wid, uint64_t ws_bits, uint64_t mean_bits) {
        const uint32_t o = HOT ? (wid & 255u) : wid;
        p.out_ws[o] = ws_bits; p.out_mean[o] = mean_bits;
    };
    auto flush = [&](bool all) {   // staged ring -> global, 16 B per lane on even slots; keeps a tail < 2 * 64 unless `all`
        if (!OSTG) return;
        lds_order();
        while (stg_n >= FL || (all && stg_n > 0)) {
            int take = stg_n >= FL ? FL : stg_n;
            const int odd = (int)(stg_w0 & 1u);
            if (odd) {
                if (lane == 0) store_pair(stg_w0, s_out[stg_w0 & (STG - 1)], s_out[STG + (stg_w0 & (STG - 1))]);
                stg_w0++; stg_n--; take--;
                if (take == 0) continue;
            }
            const int npair = take >> 1;   // <= FL / 2
            // one output column after the other: each a contiguous burst of up to FL * 8 bytes
            for (int q = lane; q < npair; q += 64) {
                const uint32_t w = stg_w0 + 2 * q;
                const uint32_t i0 = w & (STG - 1), i1 = (w + 1) & (STG - 1);
                const uint32_t o = HOT ? (w & 255u) : w;
                *reinterpret_cast<ulonglong2 *>(&p.out_ws[o]) = make_ulonglong2(s_out[i0], s_out[i1]);
            }
            for (int q = lane; q < npair; q += 64) {
                const uint32_t w = stg_w0 + 2 * q;
                const uint32_t i0 = w & (STG - 1), i1 = (w + 1) & (STG - 1);
                const uint32_t o = HOT ? (w & 255u) : w;
                *reinterpret_cast<ulonglong2 *>(&p.out_mean[o]) = make_ulonglong2(s_out[STG + i0], s_out[STG + i1]);
            }
            stg_w0 += 2 * npair; stg_n -= 2 * npair;
            if ((take & 1) && all) {
                if (lane == 0) store_pair(stg_w0, s_out[stg_w0 & (STG - 1)], s_out[STG + (stg_w0 & (STG - 1))]);
                stg_w0++; stg_n--;
            } else if (take & 1) break;
        }
        lds_order();
    };

    auto process = [&](int it, uint64_t (&ta)[CH], uint64_t (&tb)[CH], uint64_t (&va)[CH], uint64_t (&vb)[CH]) {
        const int64_t base = rbase + (int64_t)it * TILE;
        const int nloc = base + TILE <= n ? TILE : (int)(n - base);
        const uint32_t w_first = mdiv32((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ta[0]) - s0_lo, p.m32, p.sh1, p.sh2);
        int nseg = 0;
        bool unsorted = false, sat = false;
        lds_order();   // the previous chunk's walk is done with s_val / s_seg
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int l = j * 128 + 2 * lane;
            const bool pa = l < nloc, pb = l + 1 < nloc;
            const int64_t tsa = (int64_t)ta[j], tsb = (int64_t)tb[j];
            const uint32_t plo = left32((uint32_t)tb[j], (uint32_t)left_ts);
            const uint32_t phi = left32((uint32_t)(tb[j] >> 32), (uint32_t)((uint64_t)left_ts >> 32));
            const int64_t prev_ts = (int64_t)(((uint64_t)phi << 32) | plo);
            unsorted |= (pa && prev_ts > tsa) || (pb && tsa > tsb);
            const uint32_t wa = mdiv32((uint32_t)tsa - s0_lo, p.m32, p.sh1, p.sh2);
            const uint32_t wb = mdiv32((uint32_t)tsb - s0_lo, p.m32, p.sh1, p.sh2);
            const uint32_t wprev = left32(wb, left_w);
            const bool ha = pa && (wa != wprev);
            const bool hb = pb && (wb != wa);
            const uint32_t la = wa - w_first, lb = wb - w_first;
            sat |= (ha && la >= 0xFFFFu) || (hb && lb >= 0xFFFFu);
            const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
            int pos = nseg;
            pos += __builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0));
            pos += __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0));
            if (ha) s_seg[pos] = (uint32_t)l | (la << 16);
            pos += ha ? 1 : 0;
            if (hb) s_seg[pos] = (uint32_t)(l + 1) | (lb << 16);
            nseg += __popcll(ma) + __popcll(mb);
            left_w = (uint32_t)__builtin_amdgcn_readlane((int)wb, 63);
            left_ts = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(tb[j] >> 32), 63) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tb[j], 63));
            *reinterpret_cast<ulonglong2 *>(&s_val[l]) = make_ulonglong2(va[j], vb[j]);
        }
        if (nloc < TILE) {   // the data ends in this chunk: left_* are not used again
        }
        bad |= __ballot(unsorted || sat) != 0ull;
        lds_order();
        // virtual segment v: 0 = rows before the first head (they continue the carried window), v >= 1 = head v - 1
        for (int v0 = 0; v0 <= nseg; v0 += 64) {
            const int v = v0 + lane;
            const bool act = v <= nseg;
            double sum = 0.0;
            int cnt = 0;
            uint32_t wid = c_wid;
            int r0 = 0, r1 = 0;
            if (act) {
                if (v > 0) { const uint32_t e = s_seg[v - 1]; r0 = (int)(e & 0xFFFFu); wid = w_first + (e >> 16); }
                else { sum = c_sum; cnt = c_cnt; }
                r1 = v < nseg ? (int)(s_seg[v] & 0xFFFFu) : nloc;
                int r = r0;
                for (; r + 4 <= r1; r += 4) {
                    const uint64_t x0 = s_val[r], x1 = s_val[r + 1], x2 = s_val[r + 2], x3 = s_val[r + 3];
                    sum += __longlong_as_double((long long)x0); sum += __longlong_as_double((long long)x1);
                    sum += __longlong_as_double((long long)x2); sum += __longlong_as_double((long long)x3);
                }
                for (; r < r1; r++) sum += __longlong_as_double((long long)s_val[r]);
                cnt += r1 - r0;
            }
            const bool closes = act && v < nseg && (v > 0 || c_open) && wid < W32;
            const uint64_t mean_bits = (uint64_t)__double_as_longlong(sum / (double)(int64_t)cnt);
            const uint64_t ws_bits = (uint64_t)(p.s0 + (int64_t)((uint64_t)wid * (uint64_t)(uint32_t)p.interval));
            if (!OSTG && DEFER && v0 == 0) {
                // stores are in-order with loads on the wave's memory counter: issued here they would sit in front of the next
                // prefetch batch and its data would wait for their acknowledgement.  Keep the results, store them after that batch.
                pend = closes; pend_wid = wid; pend_ws = ws_bits; pend_mean = mean_bits;
            } else if (!OSTG) {
                if (closes) store_pair(wid, ws_bits, mean_bits);
            } else {
                // results of one trip are consecutive slots when no window is empty (checked); else direct stores
                const unsigned long long cm = __ballot(closes);
                if (cm) {
                    const int first = __builtin_ctzll(cm);
                    const uint32_t wq = (uint32_t)__builtin_amdgcn_readlane((int)wid, first);
                    const bool dense = __ballot(closes && wid != wq + (uint32_t)(lane - first)) == 0ull && (cm >> first) == (~0ull >> (64 - __popcll(cm)));
                    if (dense && (stg_n == 0 || stg_w0 + (uint32_t)stg_n == wq) && stg_n + __popcll(cm) <= STG) {
                        if (stg_n == 0) stg_w0 = wq;
                        if (closes) { s_out[wid & (STG - 1)] = ws_bits; s_out[STG + (wid & (STG - 1))] = mean_bits; }
                        stg_n += __popcll(cm);
                        if (stg_n >= FL) flush(false);
                    } else {
                        flush(true);
                        if (closes) store_pair(wid, ws_bits, mean_bits);
                    }
                }
            }
            // the last virtual segment stays open: it is the new carry
            if (nseg >= v0 && nseg < v0 + 64) {
                const int src = nseg - v0;
                const bool opens = nseg > 0 || c_open;
                const uint32_t slo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)__double_as_longlong(sum), src);
                const uint32_t shi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)__double_as_longlong(sum) >> 32), src);
                c_sum = __longlong_as_double((long long)(((uint64_t)shi << 32) | slo));
                c_cnt = __builtin_amdgcn_readlane(cnt, src);
                c_wid = (uint32_t)__builtin_amdgcn_readlane((int)wid, src);
                c_open = opens;
            }
        }
    };

    int it = 0;
    if (nit > 0) {
        load_set(0, ta0, tb0, va0, vb0);
        while (true) {
            if (it + 1 < nit) load_set(it + 1, ta1, tb1, va1, vb1);
            if (DEFER && pend) { store_pair(pend_wid, pend_ws, pend_mean); pend = false; }
            process(it, ta0, tb0, va0, vb0);
            it++;
            if (it >= nit) break;
            if (it + 1 < nit) load_set(it + 1, ta0, tb0, va0, vb0);
            if (DEFER && pend) { store_pair(pend_wid, pend_ws, pend_mean); pend = false; }
            process(it, ta1, tb1, va1, vb1);
            it++;
            if (it >= nit) break;
        }
        if (DEFER && pend) { store_pair(pend_wid, pend_ws, pend_mean); pend = false; }
    }
    // the window open at the end of the range: read on (128 rows at a time) until its last row
    int64_t pos = rend;
    while (c_open && pos < n) {
        const int64_t r = pos + 2 * lane;
        const bool pa = r < n, pb = r + 1 < n;
        int64_t rr = r;
        if (rr + 2 > n) rr = n - 2 > 0 ? (n - 2) & ~1ll : 0;
        const ulonglong2 t = *reinterpret_cast<const ulonglong2 *>(tsq + rr);
        const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(p.val + rr);
        const uint32_t wa = mdiv32((uint32_t)t.x - s0_lo, p.m32, p.sh1, p.sh2), wb = mdiv32((uint32_t)t.y - s0_lo, p.m32, p.sh1, p.sh2);
        lds_order();
        *reinterpret_cast<ulonglong2 *>(&s_val[2 * lane]) = x;
        lds_order();
        const unsigned long long da = __ballot(pa && wa != c_wid), db = __ballot(pb && wb != c_wid);
        // first row (in row order) outside the carried window
        int k = 128;
        if (da | db) {
            const int fa = da ? 2 * __builtin_ctzll(da) : 128, fb = db ? 2 * __builtin_ctzll(db) + 1 : 128;
            k = fa < fb ? fa : fb;
        }
        const int lim = (int)((n - pos) < 128 ? (n - pos) : 128);
        const int take = k < lim ? k : lim;
        if (lane == 0) {   // (uniform values: every lane computes the same)
        }
        double sum = c_sum;
        for (int q = 0; q < take; q++) sum += __longlong_as_double((long long)s_val[q]);
        c_sum = sum; c_cnt += take;
        pos += take;
        if (k < lim) break;   // closed by a row of the next window
    }
    if (c_open && c_wid < W32) {
        const uint64_t mean_bits = (uint64_t)__double_as_longlong(c_sum / (double)(int64_t)c_cnt);
        const uint64_t ws_bits = (uint64_t)(p.s0 + (int64_t)((uint64_t)c_wid * (uint64_t)(uint32_t)p.interval));
        if (OSTG && stg_n > 0 && stg_w0 + (uint32_t)stg_n == c_wid && stg_n < STG) {
            if (lane == 0) { s_out[c_wid & (STG - 1)] = ws_bits; s_out[STG + (c_wid & (STG - 1))] = mean_bits; }
            stg_n++;
        } else {
            flush(true);
            if (lane == 0) store_pair(c_wid, ws_bits, mean_bits);
        }
    }
    flush(true);
    if (bad && lane == 0) atomicOr(&p.status[0], 1u);
}

template <int KT, int HOT, int OSTG, int NTL, int DEFER = 0, int STG = 256>
static void launch_stream(const P &p, hipStream_t st) {
    const int64_t rows = (int64_t)KT * 512;
    const int64_t nr = (p.n + rows - 1) / rows;
    const int64_t per_xcd = (nr + 7) / 8;
    hipLaunchKernelGGL((stream_kernel<HOT, OSTG, NTL, DEFER, STG>), dim3((unsigned)(per_xcd * 8)), dim3(64), 0, st, p, nr, per_xcd, KT);
}

// ---------------------------------------------------------------------------------------------------------------
// The probe with TEMPORALLY CLUSTERED writes: one wavefront walks KT consecutive 512-row tiles (the next tile's loads in flight
// while the current one is folded), stages the dense shape's outputs (51.2 slots per tile and column) in an LDS ring and flushes
// FL slots per column at a time - FL * 8 contiguous bytes, 16 B per lane, one column after the other.  No arithmetic but the xor:
// what the write pattern alone is worth.
template <int NT, int FL>
__global__ __launch_bounds__(64) void rw_burst_kernel(const uint64_t *__restrict__ a, const uint64_t *__restrict__ b, uint64_t *__restrict__ o0,
                                                      uint64_t *__restrict__ o1, const int64_t n, const int64_t nranges, const int64_t ranges_per_xcd, const int KT) {
    constexpr int STG = 2 * FL;
    __shared__ uint64_t ring[2][STG];
    const int64_t bb = blockIdx.x;
    const int64_t rg = (bb & 7) * ranges_per_xcd + (bb >> 3);
    if (rg >= nranges) return;
    const int lane = threadIdx.x;
    const int64_t rbase = rg * (int64_t)KT * 512;
    const int nit = (int)((((rbase + (int64_t)KT * 512 < n ? rbase + (int64_t)KT * 512 : n) - rbase) / 512));
    ulonglong2 A0[4], B0[4], A1[4], B1[4];
    auto load_set = [&](int it, ulonglong2 (&A)[4], ulonglong2 (&B)[4]) {
        const ulonglong2 *pa = reinterpret_cast<const ulonglong2 *>(a + rbase + (int64_t)it * 512) + lane, *pb = reinterpret_cast<const ulonglong2 *>(b + rbase + (int64_t)it * 512) + lane;
#pragma unroll
        for (int j = 0; j < 4; j++) { A[j] = NT ? nt_load(pa + 64 * j) : pa[64 * j]; B[j] = NT ? nt_load(pb + 64 * j) : pb[64 * j]; }
    };
    int64_t stg_w0 = (rbase + 9) / 10;
    int stg_n = 0;
    auto flush = [&](bool all) {
        lds_order();
        while (stg_n >= FL || (all && stg_n > 0)) {
            int take = stg_n >= FL ? FL : stg_n;
            if (stg_w0 & 1) {
                if (lane == 0) { o0[stg_w0] = ring[0][stg_w0 & (STG - 1)]; o1[stg_w0] = ring[1][stg_w0 & (STG - 1)]; }
                stg_w0++; stg_n--; take--;
                if (take == 0) continue;
            }
            const int npair = take >> 1;
            for (int q = lane; q < npair; q += 64) {
                const int64_t w = stg_w0 + 2 * q;
                *reinterpret_cast<ulonglong2 *>(&o0[w]) = make_ulonglong2(ring[0][w & (STG - 1)], ring[0][(w + 1) & (STG - 1)]);
            }
            for (int q = lane; q < npair; q += 64) {
                const int64_t w = stg_w0 + 2 * q;
                *reinterpret_cast<ulonglong2 *>(&o1[w]) = make_ulonglong2(ring[1][w & (STG - 1)], ring[1][(w + 1) & (STG - 1)]);
            }
            stg_w0 += 2 * npair; stg_n -= 2 * npair;
            if ((take & 1) && all) {
                if (lane == 0) { o0[stg_w0] = ring[0][stg_w0 & (STG - 1)]; o1[stg_w0] = ring[1][stg_w0 & (STG - 1)]; }
                stg_w0++; stg_n--;
            } else if (take & 1) break;
        }
        lds_order();
    };
    auto process = [&](int it, ulonglong2 (&A)[4], ulonglong2 (&B)[4]) {
        const uint64_t x = A[0].x ^ A[0].y ^ A[1].x ^ A[1].y ^ A[2].x ^ A[2].y ^ A[3].x ^ A[3].y;
        const uint64_t y = B[0].x ^ B[0].y ^ B[1].x ^ B[1].y ^ B[2].x ^ B[2].y ^ B[3].x ^ B[3].y;
        const int64_t base = rbase + (int64_t)it * 512;
        const int64_t slot0 = (base + 9) / 10, slot1 = (base + 512 + 9) / 10;
        if (slot0 + lane < slot1) { ring[0][(slot0 + lane) & (STG - 1)] = x; ring[1][(slot0 + lane) & (STG - 1)] = y; }
        stg_n += (int)(slot1 - slot0);
        if (stg_n >= FL) flush(false);
    };
    int it = 0;
    if (nit > 0) {
        load_set(0, A0, B0);
        while (true) {
            if (it + 1 < nit) load_set(it + 1, A1, B1);
            process(it, A0, B0);
            if (++it >= nit) break;
            if (it + 1 < nit) load_set(it + 1, A0, B0);
            process(it, A1, B1);
            if (++it >= nit) break;
        }
    }
    flush(true);
}
template <int KT, int NT, int FL>
static void launch_rw_burst(const P &p, hipStream_t st) {
    const int64_t rows = (int64_t)KT * 512;
    const int64_t nr = (p.n + rows - 1) / rows;
    const int64_t per_xcd = (nr + 7) / 8;
    hipLaunchKernelGGL((rw_burst_kernel<NT, FL>), dim3((unsigned)(per_xcd * 8)), dim3(64), 0, st, reinterpret_cast<const uint64_t *>(p.ts), p.val, p.out_ws,
                       p.out_mean, p.n, nr, per_xcd, KT);
}

__global__ __launch_bounds__(256) void ws_fill_kernel(uint64_t *out, int64_t W, int64_t s0, int64_t interval) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 2;
    for (int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; k < W; k += stride) {
        const uint64_t a = (uint64_t)(s0 + k * interval), b = (uint64_t)(s0 + (k + 1) * interval);
        if (k + 1 < W) *reinterpret_cast<ulonglong2 *>(&out[k]) = make_ulonglong2(a, b); else out[k] = a;
    }
}
static void launch_ws_fill(const P &p, hipStream_t st) {
    hipLaunchKernelGGL(ws_fill_kernel, dim3(2048), dim3(256), 0, st, p.out_ws, p.W, p.s0, p.interval);
}
template <int TILE, int NT, int ALIGN>
static void launch_tile_split(const P &p, hipStream_t st);

__global__ void gen_kernel(int64_t n, int64_t *ts, double *val) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        ts[i] = i;
        uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + 42;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        val[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0);
    }
}
__global__ void checksum_kernel(const uint64_t *p, int64_t n, unsigned long long *out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long x = 0, s = 0;
    for (; i < n; i += stride) { uint64_t z = p[i] * 0x9E3779B97F4A7C15ull; z ^= z >> 29; x ^= z; s += z + (uint64_t)i * p[i]; }
    atomicXor(&out[0], x); atomicAdd(&out[1], s);
}

struct Variant {
    std::string name;
    void (*launch)(const P &, hipStream_t);
    bool has_output;
    std::vector<float> ms;
};

template <int TILE, int VDMA, int ABL, int OST, int NT, int WPB, int ALIGN = 0, int NTS = 0>
static void launch_tile(const P &p, hipStream_t st) {
    const int64_t ntiles = (p.n + TILE - 1) / TILE;
    const int64_t per_xcd = (ntiles + 7) / 8;
    const int64_t waves = per_xcd * 8;
    hipLaunchKernelGGL((tile_kernel<TILE, VDMA, ABL, OST, NT, WPB, ALIGN, NTS>), dim3((unsigned)((waves + WPB - 1) / WPB)), dim3(64 * WPB), 0, st, p, ntiles, per_xcd);
}
template <int TILE, int NT, int ALIGN>
static void launch_tile_split(const P &p, hipStream_t st) {   // the window starts by a fill kernel, the means by the tile kernel
    launch_ws_fill(p, st);
    launch_tile<TILE, 0, 4, 0, NT, 1, ALIGN, 5>(p, st);
}
template <int MODE, int NT, int WPB, int ST = 0>
static void launch_rw(const P &p, hipStream_t st) {
    const int64_t ntiles = p.n / 512;
    const int64_t per_xcd = (ntiles + 7) / 8;
    const int64_t waves = per_xcd * 8;
    hipLaunchKernelGGL((rw_ceiling_kernel<MODE, NT, WPB, ST>), dim3((unsigned)((waves + WPB - 1) / WPB)), dim3(64 * WPB), 0, st,
                       reinterpret_cast<const uint64_t *>(p.ts), p.val, p.out_ws, p.out_mean, ntiles, per_xcd);
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? (int64_t)atof(argv[1]) : 1000000000ll;
    const int rounds = argc > 2 ? atoi(argv[2]) : 7;
    const char *only = argc > 3 ? argv[3] : nullptr;
    const int64_t interval = 10;
    const int64_t W = (n + interval - 1) / interval;
    int64_t *ts; double *val; uint64_t *o0, *o1; uint32_t *status; unsigned long long *cks;
    CK(hipMalloc(&ts, n * 8 + 64)); CK(hipMalloc(&val, n * 8 + 64));
    CK(hipMalloc(&o0, W * 8 + 64)); CK(hipMalloc(&o1, W * 8 + 64));
    CK(hipMalloc(&status, 256)); CK(hipMalloc(&cks, 16));
    hipLaunchKernelGGL(gen_kernel, dim3(256 * 16), dim3(256), 0, 0, n, ts, val);
    CK(hipDeviceSynchronize());
    P p;
    p.ts = ts; p.val = reinterpret_cast<const uint64_t *>(val); p.n = n; p.s0 = 0; p.interval = interval; p.W = W;
    { const uint64_t d = (uint64_t)interval; int l = 0; while ((1ull << l) < d) l++;
      p.m32 = (uint32_t)((((1ull << l) - d) << 32) / d) + 1; p.sh1 = l < 1 ? l : 1; p.sh2 = l > 1 ? l - 1 : 0; }
    p.out_ws = o0; p.out_mean = o1; p.status = status;

    std::vector<Variant> vs;
#define V(name, fn, out) vs.push_back(Variant{name, fn, out, {}})
    V("t512_reg_full", (launch_tile<512, 0, 4, 0, 0, 1>), true);
    V("t512_reg_al16_nti", (launch_tile<512, 0, 4, 0, 2, 1, 16>), true);
    V("t512_reg_al16_nti_nts", (launch_tile<512, 0, 4, 0, 2, 1, 16, 1>), true);
    V("t512_reg_nti_nts", (launch_tile<512, 0, 4, 0, 2, 1, 0, 1>), true);
    V("t512_reg_sc1", (launch_tile<512, 0, 4, 0, 0, 1, 0, 4>), true);
    V("t512_reg_nti_sc1", (launch_tile<512, 0, 4, 0, 2, 1, 0, 4>), true);
    V("t1024_reg_al16_nti", (launch_tile<1024, 0, 4, 0, 2, 1, 16>), true);
    V("t1024_reg_al16_nti_nts", (launch_tile<1024, 0, 4, 0, 2, 1, 16, 1>), true);
    V("t1024_dma_al16_nti_nts", (launch_tile<1024, 1, 4, 0, 2, 1, 16, 1>), true);
    V("t512_meanonly", (launch_tile<512, 0, 4, 0, 0, 1, 0, 5>), false);
    V("t512_meanonly_nti", (launch_tile<512, 0, 4, 0, 2, 1, 0, 5>), false);
    V("ws_fill", launch_ws_fill, false);
    V("t512_split", (launch_tile_split<512, 0, 0>), true);
    V("t512_split_nti_al16", (launch_tile_split<512, 2, 16>), true);
    V("t1024_split_nti_al16", (launch_tile_split<1024, 2, 16>), true);
    V("t512_reg_hotst", (launch_tile<512, 0, 4, 0, 0, 1, 0, 2>), false);
    V("rw_reg_st0", (launch_rw<0, 0, 1, 0>), false);
    V("rw_regnt_st0", (launch_rw<0, 1, 1, 0>), false);
    V("rw_reg_st2_al16", (launch_rw<0, 0, 1, 2>), false);
    V("rw_reg_st7_lds", (launch_rw<0, 0, 1, 7>), false);
    V("rw_reg_st4_none", (launch_rw<0, 0, 1, 4>), false);
    V("rw_regnt_st4_none", (launch_rw<0, 1, 1, 4>), false);
    // round 3: temporally clustered writes (K consecutive tiles per wavefront, bursts of FL slots = FL * 8 bytes per output column)
    V("burst_rw_k8_1k", (launch_rw_burst<8, 1, 128>), false);
    V("burst_rw_k8_2k", (launch_rw_burst<8, 1, 256>), false);
    V("burst_rw_k16_4k", (launch_rw_burst<16, 1, 512>), false);
    V("burst_rw_k32_8k", (launch_rw_burst<32, 1, 1024>), false);
    V("burst_stream_k4_1k", (launch_stream<4, 0, 1, 1, 0, 256>), true);
    V("burst_stream_k8_2k", (launch_stream<8, 0, 1, 1, 0, 512>), true);
    V("burst_stream_k16_4k", (launch_stream<16, 0, 1, 1, 0, 1024>), true);
    V("burst_stream_k16_direct", (launch_stream<16, 0, 0, 1, 0, 256>), true);
    if (only) {   // comma-separated substrings
        std::vector<std::string> pats;
        std::string o(only);
        size_t a = 0;
        while (a <= o.size()) { size_t b = o.find(',', a); if (b == std::string::npos) b = o.size(); if (b > a) pats.push_back(o.substr(a, b - a)); a = b + 1; }
        vs.erase(std::remove_if(vs.begin(), vs.end(), [&](const Variant &v) {
            for (auto &q : pats) if (v.name.find(q) != std::string::npos) return false;
            return true; }), vs.end());
    }

    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // correctness: checksums of both outputs per variant
    unsigned long long ref[4] = {0, 0, 0, 0};
    bool have_ref = false;
    for (auto &v : vs) {
        if (!v.has_output) continue;
        CK(hipMemsetAsync(o0, 0xAB, W * 8, st)); CK(hipMemsetAsync(o1, 0xAB, W * 8, st)); CK(hipMemsetAsync(status, 0, 256, st));
        v.launch(p, st);
        unsigned long long h[4];
        CK(hipMemsetAsync(cks, 0, 16, st));
        hipLaunchKernelGGL(checksum_kernel, dim3(4096), dim3(256), 0, st, o0, W, cks);
        CK(hipMemcpyAsync(h, cks, 16, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        CK(hipMemsetAsync(cks, 0, 16, st));
        hipLaunchKernelGGL(checksum_kernel, dim3(4096), dim3(256), 0, st, o1, W, cks);
        CK(hipMemcpyAsync(h + 2, cks, 16, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        uint32_t hs[8];
        CK(hipMemcpy(hs, status, 32, hipMemcpyDeviceToHost));
        CK(hipGetLastError());
        if (!have_ref) { memcpy(ref, h, sizeof ref); have_ref = true; }
        const bool ok = memcmp(ref, h, sizeof ref) == 0 && hs[0] == 0 && hs[4] == 0;
        printf("check %-24s %s  (%016llx %016llx)\n", v.name.c_str(), ok ? "ok" : "MISMATCH", h[0], h[2]);
        fflush(stdout);
    }
    // spot values on the host: window 0 and the last window
    {
        double m0; int64_t w0;
        CK(hipMemcpy(&m0, o1, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&w0, o0 + (W - 1), 8, hipMemcpyDeviceToHost));
        printf("mean[0] = %.17g, window_start[W-1] = %lld (expect %lld)\n", m0, (long long)w0, (long long)((W - 1) * interval));
    }
    for (int r = 0; r < rounds + 1; r++)
        for (auto &v : vs) {
            CK(hipEventRecord(e0, st));
            v.launch(p, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) v.ms.push_back(ms);
        }
    CK(hipGetLastError());
    printf("\n%-26s %9s %9s %9s   %s\n", "variant", "min ms", "median", "max", "read TB/s @16 B/row (median) | frac of 8");
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2];
        const double tbs = (double)n * 16.0 / (med * 1e-3) / 1e12;
        printf("%-26s %9.4f %9.4f %9.4f   %.3f | %.3f\n", v.name.c_str(), v.ms.front(), med, v.ms.back(), tbs, tbs / 8.0);
    }
    return 0;
}
