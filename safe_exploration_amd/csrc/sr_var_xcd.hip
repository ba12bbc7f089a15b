// sr_var_xcd.hip -- K2x: the variance contraction for FEW query tiles (128 < T <= 1024 on models of a few thousand points),
// with the k range partitioned by XCD.
//
//   part[d][rb][t] = sum_{i in row block rb} ( sum_{k <= i} Wt[d][k][i] K*[d][k][t] )^2
//
// With one to eight query tiles every 128 x 128 block of U^-1 is used once per query tile, the arithmetic intensity of a
// (row block, k-block) cell is 16 flop per byte if both of its operand blocks come from memory -- on the ridge of the chip
// -- and 32 if the K* block comes from the L2.  The balanced-share kernel K2b (sr_predict.hip) walks the cells row by row:
// every workgroup touches its own k range of K*, eight non-coherent L2s each see all of K* while 26 MB of U^-1 stream
// through them, and 83 % of the L2 requests miss (profiles/r03: N = 5000, T = 128: 412 MB over the fabric for 215 MB of
// unique operands, MFMA busy 0.64).  Here
//   * the cells of all outputs are cut into eight SLABS of equal area along k (column-major over (output, k): a slab is a
//     k range of one output, or the tail of one output and the head of the next), one slab per XCD (workgroup b runs on
//     XCD b % 8 -- an observation the speed relies on, never the result): an XCD reads ITS rows of K* and nothing else
//     of it, and they stay in its L2 (<= 2.6 MB at N = 5000, T = 128);
//   * inside a slab the segments (row block x slab) are listed heavy rows first and cut into equal contiguous shares, one
//     per GROUP of workgroups; the ntq workgroups of a group take the same share for the ntq query tiles side by side
//     on the same XCD, so the U^-1 blocks of a share cross the fabric once and serve all query tiles from the L2;
//   * one workgroup per CU with FOUR LDS stages (144 KB): three k-tiles of LDS-DMA in flight per workgroup cover the
//     latency of operands that really come from HBM (the two-stage loop of the big-batch kernel has one in flight and
//     relies on the second workgroup of the CU);
//   * a run that is not a whole tile leaves its 128 x 128 partial product in a slot that follows from the geometry alone
//     (first / last run of a share: the share's two slots; a whole segment inside a share: the slot of (tile, slab)), and
//     sr_var_xcd_reduce_kernel adds a tile's partial products in ascending k order: deterministic, whoever ran first.
// Caller to match: the pool scoring of ssm_gpy/gaussian_process.py:333 and the particle batches of
// sampling_models.py:66-80 (a few hundred queries against the full model).
#include "sr_mfma_tile.h"

namespace {

constexpr int XCDS = 8;
constexpr int MAXP = 4;                      // slab pieces per XCD (n_out <= 8: at most 2)
constexpr int MAXRUN = 12;                   // runs of one tile inside one slab (host-checked)
constexpr int UPB = srt::BM / srt::BK;       // k units (of BK rows) per 128-block: 8
constexpr long TILE = (long)srt::BM * srt::BN;

// geometry of one launch, by value in the kernel arguments (host: xcd_geometry)
struct Geo {
    int nrb, ntq, ub, n_out;                 // row blocks, query tiles, first real k unit (front padding), outputs
    int ngrp;                                // groups (shares) per XCD
    int npiece[XCDS];
    int pd[XCDS][MAXP], pa[XCDS][MAXP], pb[XCDS][MAXP];      // piece: output, k units [pa, pb)
    int L[XCDS];                             // total segment length of the XCD's list (k units); L * ngrp < 2^31
};

// length (k units) of the segment (piece [va, vb), row block rb): the part of the piece with k < 128 (rb + 1)
__host__ __device__ inline int seg_len(int va, int vb, int rb) {
    const int e = min(vb, UPB * (rb + 1));
    return e > va ? e - va : 0;
}
// sum of seg_len over the row blocks r_from .. nrb - 1 of a piece, in closed form (rows from r_full on hold the full
// width vb - va, the rows below end on the diagonal: 8 (r + 1) - va)
__host__ __device__ inline int seg_tail_sum(int va, int vb, int r_from, int nrb) {
    const int r_lo = va / UPB;
    const int r_full = (vb + UPB - 1) / UPB - 1;
    const int a = r_from > r_lo ? r_from : r_lo;
    if (a >= nrb) return 0;
    const int f0 = a > r_full ? a : r_full;
    int sum = (nrb - f0 > 0 ? nrb - f0 : 0) * (vb - va);
    const int d1 = (r_full < nrb ? r_full : nrb) - 1;          // last diagonal row
    if (d1 >= a) {
        const int cnt = d1 - a + 1;
        sum += UPB * ((a + 1 + d1 + 1) * cnt / 2) - cnt * va;
    }
    return sum;
}
__host__ __device__ inline int share_bound(int g, int L, int n) { return (int)(((unsigned)g * (unsigned)L) / (unsigned)n); }
__host__ __device__ inline int share_owner(int u, int L, int n) {              // g with bound(g) <= u < bound(g + 1)
    int g = (int)(((unsigned)u * (unsigned)n) / (unsigned)L);
    while (g + 1 <= n && share_bound(g + 1, L, n) <= u) ++g;
    while (g > 0 && share_bound(g, L, n) > u) --g;
    return g;
}

// slot of a run [r0, r1) (positions in XCD c's list) of group g (share [s0, s1)) of tile (d, rb), query tile x
__device__ __forceinline__ long run_slot(const Geo& G, int c, int g, int x, int r0, int r1, int s0, int s1, int d, int rb) {
    const long nA = 2L * XCDS * G.ngrp * G.ntq;
    if (r0 == s0) return ((((long)c * G.ngrp + g) * G.ntq + x) * 2 + 0);
    if (r1 == s1) return ((((long)c * G.ngrp + g) * G.ntq + x) * 2 + 1);
    return nA + ((((long)d * G.nrb + rb) * G.ntq + x) * XCDS + c);            // the whole segment, strictly inside the share
}

// the runs of one share, in list order
struct Run { int d, rb, k0, k1, r0, r1; bool valid; };
struct RunIter {
    int c, s0, s1, p, rb, pos;
    __device__ __forceinline__ void start(const Geo& G, int c_, int s0_, int s1_) {
        c = c_; s0 = s0_; s1 = s1_; p = 0; pos = 0;
        rb = G.npiece[c] > 0 ? G.nrb - 1 : -1;
    }
    __device__ __forceinline__ Run next(const Geo& G) {
        Run r{};
        r.valid = false;
        while (p < G.npiece[c] && pos < s1) {
            const int va = G.pa[c][p], vb = G.pb[c][p];
            if (rb < va / UPB) { ++p; rb = G.nrb - 1; continue; }
            const int len = seg_len(va, vb, rb);
            const int at = pos, row = rb;
            pos += len; --rb;
            if (len > 0 && at + len > s0 && at < s1) {
                r.r0 = at > s0 ? at : s0;
                r.r1 = at + len < s1 ? at + len : s1;
                r.d = G.pd[c][p]; r.rb = row;
                r.k0 = (va + (r.r0 - at)) * srt::BK; r.k1 = (va + (r.r1 - at)) * srt::BK;
                r.valid = true;
                return r;
            }
        }
        return r;
    }
};

// ---- main loop: acc += A[k][m] B[k][n] over [k_beg, k_end), NS LDS stages, NS - 1 k-tiles of LDS-DMA in flight -----------
// NW wavefronts per workgroup: 4 (2 x 2 of 64 x 64) or 8 (2 x 4 of 64 x 32: two wavefronts per SIMD from ONE workgroup)
template <int NW> struct AccX {
    static constexpr int NI = NW == 8 ? 2 : 4;
    d4_t v[4][NI];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) v[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
    }
};

template <int NS, int NW>
__device__ __forceinline__ void wait_then_barrier(int inflight) {             // stages that may stay in flight
    constexpr int PER = 2 * srt::BK / NW;                                     // DMA instructions per stage and wavefront
    if (NS >= 4 && inflight >= 2) {
        if (PER == 8) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    } else if (NS >= 3 && inflight == 1) {
        if (PER == 8) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int NS, int NW, int DBG>
struct Loop {
    static constexpr int NI = AccX<NW>::NI;
    static constexpr int WN = NW == 8 ? 4 : 2;       // wavefront columns
    static constexpr int STG = srt::BK * srt::LDT;
    const double* ga; const double* gb; long lda, ldb;
    double *As, *Bs;
    int srow;
    __device__ __forceinline__ void bind(const double* A, long lda_, const double* B, long ldb_, double* smem) {
        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        lda = lda_; ldb = ldb_;
        ga = A + (long)wave * lda + 2 * lane;            // rows wave, wave + NW, .. of a k-tile
        gb = B + (long)wave * ldb + 2 * lane;
        As = smem; Bs = smem + NS * STG;
        srow = wave * srt::LDT;
    }
    __device__ __forceinline__ void dma(int k0, int buf) const {
        if (DBG == 2) return;
        const double* pa_ = ga + (long)k0 * lda;
        const double* pb_ = gb + (long)k0 * ldb;
        double* sa_ = As + buf * STG + srow;
        double* sb_ = Bs + buf * STG + srow;
#pragma unroll
        for (int j_ = 0; j_ < srt::BK / NW; ++j_) {
            __builtin_amdgcn_global_load_lds(SRT_AS1(pa_ + NW * j_ * lda), SRT_AS3(sa_ + NW * j_ * srt::LDT), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(SRT_AS1(pb_ + NW * j_ * ldb), SRT_AS3(sb_ + NW * j_ * srt::LDT), 16, 0, 0);
        }
    }
    // the first NS - 1 k-tiles of [k_beg, k_end) on their way (all stages must be free)
    __device__ __forceinline__ void prime(int k_beg, int k_end) const {
        const int nsteps = (k_end - k_beg) / srt::BK;
#pragma unroll
        for (int s_ = 0; s_ < NS - 1; ++s_)
            if (s_ < nsteps) dma(k_beg + s_ * srt::BK, s_);
    }
    // after prime(); `others`: this wavefront has stores in flight that were issued after the prime (they may retire out
    // of order with the loads: the first wait is for everything)
    __device__ __forceinline__ void run(int k_beg, int k_end, AccX<NW>& acc, bool others) const {
        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int wm = wave / WN, wn = wave % WN;
        const int nsteps = (k_end - k_beg) / srt::BK;
        const int fa = (lane >> 4) * srt::LDT + wm * 64 + (lane & 15);
        const int fb = (lane >> 4) * srt::LDT + wn * (16 * NI) + (lane & 15);
        int buf = 0, nbuf = NS - 1;
        for (int t = 0; t < nsteps; ++t) {
            const int issued = (t + NS - 1 < nsteps) ? t + NS - 1 : nsteps;
            wait_then_barrier<NS, NW>((t == 0 && others) ? 0 : issued - (t + 1));     // k-tile t has landed; stage nbuf is free
            if (t + NS - 1 < nsteps) dma(k_beg + (t + NS - 1) * srt::BK, nbuf);
            const double* as = As + buf * STG + fa;
            const double* bs = Bs + buf * STG + fb;
            if (DBG != 1)
#pragma unroll
            for (int kk = 0; kk < srt::BK / 4; ++kk) {
                double af[4], bf[NI];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = as[kk * 4 * srt::LDT + i * 16];
#pragma unroll
                for (int i = 0; i < NI; ++i) bf[i] = bs[kk * 4 * srt::LDT + i * 16];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc.v[i][j], 0, 0, 0);
            }
            buf = (buf + 1 == NS) ? 0 : buf + 1;
            nbuf = (nbuf + 1 == NS) ? 0 : nbuf + 1;
        }
        __syncthreads();                      // every wavefront is through with the stages: they may be primed again
    }
};

// sum over the 64 rows of a wavefront row of the squared accumulators: red[wm][128 columns], then the two rows added
template <int NW, int NMI>
__device__ __forceinline__ void square_reduce_store(const double (&vals)[4 * AccX<NW>::NI * 4], double* red,
                                                    double* dst /* 128 columns */) {
    constexpr int NI = AccX<NW>::NI;
    constexpr int WN = NW == 8 ? 4 : 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        double v = 0.0;
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
            for (int q = 0; q < 4; ++q) v = fma(vals[(mi * NI + ni) * 4 + q], vals[(mi * NI + ni) * 4 + q], v);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) red[wm * 128 + wn * (16 * NI) + ni * 16 + lane] = v;
    }
    __syncthreads();
    if (tid < 128) dst[tid] = red[tid] + red[128 + tid];
    __syncthreads();
}

// A partial product in its slot: 16-byte stores (two doubles of an accumulator quad per lane and instruction: the partial
// products are store-issue bound -- 128 KB per run from 256 or 512 lanes), element (mi, ni, half h) of lane tid at
// ((mi NI + ni) 2 + h) NT + tid double2's.
template <int NS, int NW, int DBG = 0>
__global__ __launch_bounds__(64 * NW, (NS <= 2 && NW == 4) ? 2 : 1)
void sr_var_xcd_kernel(const double* __restrict__ Wt, const double* __restrict__ Ks, double* __restrict__ Vt,
                       double* __restrict__ part, int Np, long Tp, Geo G) {
    constexpr int NT = 64 * NW, NI = AccX<NW>::NI;
    __shared__ double smem[NS * 2 * srt::BK * srt::LDT];
    __shared__ double red[256];
    const int c = blockIdx.x & (XCDS - 1);
    const int i = blockIdx.x >> 3;
    const int g = i / G.ntq;
    const int x = i % G.ntq;
    const int L = G.L[c];
    const int s0 = share_bound(g, L, G.ngrp), s1 = share_bound(g + 1, L, G.ngrp);
    const int tid = threadIdx.x;
    RunIter it;
    it.start(G, c, s0, s1);
    Run cur = it.next(G);
    Loop<NS, NW, DBG> lp;
    bool others = false;
    if (cur.valid) {
        lp.bind(Wt + (long)cur.d * Np * Np + (long)cur.rb * srt::BM, Np, Ks + (long)cur.d * Np * Tp + (long)x * srt::BN, Tp, smem);
        lp.prime(cur.k0, cur.k1);
    }
    while (cur.valid) {
        AccX<NW> acc;
        acc.zero();
        lp.run(cur.k0, cur.k1, acc, others);
        // the next run's first k-tiles start their way from memory BEFORE this run's partial product is stored
        const Run nxt = it.next(G);
        if (nxt.valid) {
            lp.bind(Wt + (long)nxt.d * Np * Np + (long)nxt.rb * srt::BM, Np, Ks + (long)nxt.d * Np * Tp + (long)x * srt::BN, Tp, smem);
            lp.prime(nxt.k0, nxt.k1);
        }
        const bool whole = (cur.k0 == G.ub * srt::BK) && (cur.k1 == (cur.rb + 1) * srt::BM);
        if (!whole && DBG == 3) {
        } else if (!whole) {
            double2* slot = reinterpret_cast<double2*>(Vt + run_slot(G, c, g, x, cur.r0, cur.r1, s0, s1, cur.d, cur.rb) * TILE);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        slot[((mi * NI + ni) * 2 + h) * NT + tid] = double2{acc.v[mi][ni][2 * h], acc.v[mi][ni][2 * h + 1]};
            others = true;
        } else {                               // the whole k range of the tile: square and reduce on the spot
            double vals[4 * NI * 4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) vals[(mi * NI + ni) * 4 + q] = acc.v[mi][ni][q];
            // four partial norms per tile (the layout of the reduce pass): the whole norm and three zeros
            double* pt = part + ((long)cur.d * 4 * G.nrb + cur.rb * 4) * Tp + (long)x * srt::BN;
            square_reduce_store<NW, 4>(vals, red, pt);
            if (tid >= 128 && tid < 256) for (int q = 1; q < 4; ++q) pt[q * Tp + tid - 128] = 0.0;
            others = true;
        }
        cur = nxt;
    }
}

// grid (4, n_out * ntq * nrb): workgroup (mi, tile) adds row mi of MFMA tiles of the tile's partial products, in ascending k.
// The slots of a tile's partial products follow from the geometry (closed forms: no list travels between the launches).
template <int NW>
__global__ __launch_bounds__(64 * NW) void sr_var_xcd_reduce_kernel(const double* __restrict__ Vt, double* __restrict__ part,
                                                                    long Tp, Geo G) {
    constexpr int NT = 64 * NW, NI = AccX<NW>::NI, NE2 = NI * 2;
    __shared__ double red[256];
    __shared__ long slots[64];
    __shared__ long sl[XCDS][MAXRUN];
    __shared__ int cnt[XCDS];
    __shared__ int nslots;
    const int mi = blockIdx.x;
    const int rb = blockIdx.y % G.nrb;
    const int dx = blockIdx.y / G.nrb;
    const int d = dx / G.ntq, x = dx % G.ntq;
    const int tid = threadIdx.x;
    if (tid < XCDS) {                          // lane c lists the runs of the tile inside slab c (closed forms)
        const int c = tid;
        int n = 0, base = 0;                   // base: position of piece p's first segment in the XCD's list
        for (int p = 0; p < G.npiece[c]; ++p) {
            const int va = G.pa[c][p], vb = G.pb[c][p];
            const int len = seg_len(va, vb, rb);
            if (G.pd[c][p] == d && len > 0) {
                const int P = base + seg_tail_sum(va, vb, rb + 1, G.nrb);       // the segments of the rows above rb
                const int L = G.L[c];
                const int ga = share_owner(P, L, G.ngrp), gb = share_owner(P + len - 1, L, G.ngrp);
                for (int g = ga; g <= gb; ++g) {
                    const int s0 = share_bound(g, L, G.ngrp), s1 = share_bound(g + 1, L, G.ngrp);
                    const int r0 = P > s0 ? P : s0, r1 = (P + len < s1) ? P + len : s1;
                    const bool whole = va + (r0 - P) == G.ub && va + (r1 - P) == UPB * (rb + 1);
                    if (n < MAXRUN) sl[c][n] = whole ? -1 : run_slot(G, c, g, x, r0, r1, s0, s1, d, rb);
                    ++n;
                }
            }
            base += seg_tail_sum(va, vb, 0, G.nrb);
        }
        cnt[c] = n < MAXRUN ? n : MAXRUN;
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        bool direct = false;
        for (int c = 0; c < XCDS; ++c)
            for (int r = 0; r < cnt[c]; ++r) {
                if (sl[c][r] < 0) direct = true;
                else if (n < 64) slots[n++] = sl[c][r];
            }
        nslots = direct ? 0 : n;
    }
    __syncthreads();
    const int nseg = nslots;
    if (nseg == 0) return;                     // finished by the one workgroup that held the whole tile
    double2 v[NE2];
#pragma unroll
    for (int e = 0; e < NE2; ++e) v[e] = double2{0.0, 0.0};
    for (int sg0 = 0; sg0 < nseg; sg0 += 4) {                   // four partial products' loads in flight
        double2 w[4][NE2];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int sg = sg0 + b;
            const double2* src = reinterpret_cast<const double2*>(Vt + (sg < nseg ? slots[sg] : slots[0]) * TILE) + (long)(mi * NE2) * NT + tid;
#pragma unroll
            for (int e = 0; e < NE2; ++e) w[b][e] = (sg < nseg) ? src[e * NT] : double2{0.0, 0.0};
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < NE2; ++e) { v[e].x += w[b][e].x; v[e].y += w[b][e].y; }
    }
    double vals[4 * NI * 4];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        vals[ni * 4 + 0] = v[ni * 2].x; vals[ni * 4 + 1] = v[ni * 2].y;
        vals[ni * 4 + 2] = v[ni * 2 + 1].x; vals[ni * 4 + 3] = v[ni * 2 + 1].y;
    }
    square_reduce_store<NW, 1>(vals, red, part + ((long)d * 4 * G.nrb + rb * 4 + mi) * Tp + (long)x * srt::BN);
}

// slabs of equal area in the column-major order over (output, k unit); false if some XCD would hold more than MAXP pieces
bool xcd_geometry(int N, int Np, long Tp, int n_out, int wgs_per_xcd, Geo* out) {
    Geo G{};
    G.nrb = Np / srt::BM; G.ntq = (int)(Tp / srt::BN); G.n_out = n_out;
    G.ub = ((Np - N) / srt::BK);
    if (G.ntq < 1 || wgs_per_xcd % G.ntq != 0) return false;
    G.ngrp = wgs_per_xcd / G.ntq;
    const int nu = UPB * G.nrb;
    auto height = [&](int v) { return (long)(G.nrb - v / UPB); };
    long tri = 0;
    for (int v = G.ub; v < nu; ++v) tri += height(v);
    const long Atot = tri * n_out;
    // boundary c: the first column (d, v) whose cumulative area reaches c Atot / 8
    int bd[XCDS + 1], bv[XCDS + 1];
    {
        long cum = 0;
        int c = 0, d = 0, v = G.ub;
        while (c <= XCDS) {
            const long target = (Atot * c) / XCDS;
            while (d < n_out && cum < target) {
                cum += height(v);
                if (++v == nu) { v = G.ub; ++d; }
            }
            bd[c] = d; bv[c] = v;
            ++c;
        }
        bd[XCDS] = n_out; bv[XCDS] = G.ub;
    }
    for (int c = 0; c < XCDS; ++c) {
        int np = 0;
        int d = bd[c], v = bv[c];
        while (d < bd[c + 1] || (d == bd[c + 1] && v < bv[c + 1])) {
            const int vend = (d == bd[c + 1]) ? bv[c + 1] : nu;
            if (vend > v) {
                if (np == MAXP) return false;
                G.pd[c][np] = d; G.pa[c][np] = v; G.pb[c][np] = vend;
                ++np;
            }
            ++d; v = G.ub;
        }
        G.npiece[c] = np;
        long L = 0;
        for (int p = 0; p < np; ++p) L += seg_tail_sum(G.pa[c][p], G.pb[c][p], 0, G.nrb);
        if (L < G.ngrp || L * (G.ngrp + 1) >= (1L << 31)) return false;   // (fewer k units than shares; 32-bit share arithmetic)
        G.L[c] = (int)L;
    }
    // the reduce pass lists at most MAXRUN partial products per (tile, slab) and 64 per tile: bound them with the widest
    // segment of every slab (the bottom row block holds the full width of every piece)
    long per_tile = 0;
    for (int c = 0; c < XCDS; ++c) {
        const long share = std::max(1, G.L[c] / G.ngrp);
        long in_slab = 0;
        for (int p = 0; p < G.npiece[c]; ++p) in_slab = std::max(in_slab, (long)seg_len(G.pa[c][p], G.pb[c][p], G.nrb - 1) / share + 2);
        if (in_slab > MAXRUN) return false;
        per_tile += in_slab;
    }
    if (per_tile > 64) return false;
    *out = G;
    return true;
}

// Which loop (SR_XCD_VARIANT; measured at N = 5000, T = 128, main kernel alone): 42 = four wavefronts, two LDS stages, TWO
// workgroups per CU: 115 us (default); 84 = eight wavefronts, four stages, one workgroup per CU: 124 us; 44 = four
// wavefronts, four stages, one per CU: 152 us.  One workgroup per CU loses although its operands arrive three k-tiles ahead:
// all wavefronts of the CU meet at the same barrier every k-tile and the matrix pipe idles through the fragment reads
// behind it, which a second, unsynchronised workgroup covers.
static int xcd_variant() {
    static const int v = getenv("SR_XCD_VARIANT") ? atoi(getenv("SR_XCD_VARIANT")) : 84;
    return v;
}
#define WGS_PER_XCD (xcd_variant() == 42 ? 64 : 32)

}  // namespace

long sr_var_xcd_ws(int Np, long Tp, int n_out) {
    const long nrb = Np / srt::BM, ntq = Tp / srt::BN;
    return (2L * XCDS * 64 + (long)n_out * nrb * ntq * XCDS) * TILE;
}

bool sr_var_xcd_wanted(int N, int Np, long Tp, int n_out) {
    const long nrb = Np / srt::BM, ntq = Tp / srt::BN;
    if (!(ntq == 1 || ntq == 2 || ntq == 4 || ntq == 8)) return false;
    if ((long)n_out * ntq * nrb * (nrb + 1) / 2 < SR_VAR_XCD_MIN_CELLS) return false;
    Geo G;
    return xcd_geometry(N, Np, Tp, n_out, WGS_PER_XCD, &G);
}

int sr_launch_var_xcd(const double* Wt, const double* Ks, double* Vt, double* part, int N, int Np, long Tp, int n_out,
                      hipStream_t s) {
    Geo G;
    SR_CHECK(xcd_geometry(N, Np, Tp, n_out, WGS_PER_XCD, &G), SR_EUNSUPPORTED, "var_xcd: no slab partition for Np=%d Tp=%ld n_out=%d",
             Np, Tp, n_out);
    const dim3 grid(XCDS * WGS_PER_XCD), rgrid(4, n_out * G.ntq * G.nrb);
    const int var = xcd_variant();
    // ablations of the main kernel (SR_XCD_DEBUG, variant 84 / 42 only; results are garbage): 1 no MFMAs, 2 no operand DMA,
    // 3 no partial-product stores -- profiles/r04_xcd_ablation.txt
    static const int dbg = getenv("SR_XCD_DEBUG") ? atoi(getenv("SR_XCD_DEBUG")) : 0;
#define SRX_LAUNCH(NS_, NW_, DBG_) hipLaunchKernelGGL((sr_var_xcd_kernel<NS_, NW_, DBG_>), grid, dim3(64 * NW_), 0, s, Wt, Ks, Vt, part, Np, Tp, G)
    if (var == 42) {
        if (dbg == 1) SRX_LAUNCH(2, 4, 1); else if (dbg == 2) SRX_LAUNCH(2, 4, 2); else if (dbg == 3) SRX_LAUNCH(2, 4, 3); else SRX_LAUNCH(2, 4, 0);
    } else if (var == 44) SRX_LAUNCH(4, 4, 0);
    else {
        if (dbg == 1) SRX_LAUNCH(4, 8, 1); else if (dbg == 2) SRX_LAUNCH(4, 8, 2); else if (dbg == 3) SRX_LAUNCH(4, 8, 3); else SRX_LAUNCH(4, 8, 0);
    }
#undef SRX_LAUNCH
    SR_HIP(hipGetLastError());
    if (var == 42 || var == 44) hipLaunchKernelGGL(sr_var_xcd_reduce_kernel<4>, rgrid, dim3(256), 0, s, Vt, part, Tp, G);
    else hipLaunchKernelGGL(sr_var_xcd_reduce_kernel<8>, rgrid, dim3(512), 0, s, Vt, part, Tp, G);
    SR_HIP(hipGetLastError());
    return SR_OK;
}
