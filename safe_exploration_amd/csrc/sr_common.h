// sr_common.h -- shared declarations for libsafereach (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cstring>
#include <vector>
#include "../../include/safereach.h"

#define SR_NB 128          // factor block size == GEMM tile edge; Np is a multiple of it
#define SR_PANEL 4         // factor blocks per Cholesky panel (deferred trailing update with K = 512)
#define SR_MAX_NS 8
#define SR_MAX_DEVICES 64    // per-device tables of the host side (locks)
#define SR_FACT_SLOTS 8      // outputs factorised concurrently (own streams + scratch each)
#define SR_MAX_NU 4
#define SR_MAX_D 12
#define SR_VAR_CLIP 1e-15  // GPy GP._raw_predict clips the predictive variance here

void sr_set_error(const char* fmt, ...);

// Measurement switches exist in the LAB build only (make lab: scripts/_bin/libsafereach_lab.so, -DSR_LAB; the scripts that
// sweep them load it through SAFEREACH_LIB).  In the product every one of them is its default, at compile time: the
// library reads no environment variable except the three allocator diagnostics of sr_capi_handle.hip.
#include <cstdlib>
#ifdef SR_LAB
static inline const char* sr_lab_str(const char* name) { return getenv(name); }
#else
static inline constexpr const char* sr_lab_str(const char*) { return nullptr; }
#endif
static inline long sr_lab_env(const char* name, long dflt) { const char* v = sr_lab_str(name); return v ? atol(v) : dflt; }
static inline double sr_lab_envf(const char* name, double dflt) { const char* v = sr_lab_str(name); return v ? atof(v) : dflt; }
static inline bool sr_lab_on(const char* name) { return sr_lab_str(name) != nullptr; }

#define SR_HIP(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            sr_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return SR_EHIP;                                                           \
        }                                                                             \
    } while (0)

#define SR_CHECK(cond, code, ...)                                                     \
    do {                                                                              \
        if (!(cond)) { sr_set_error(__VA_ARGS__); return (code); }                    \
    } while (0)

#define SR_TRY(expr)                                                                  \
    do { int rc_ = (expr); if (rc_ != SR_OK) return rc_; } while (0)

typedef double d4_t __attribute__((ext_vector_type(4)));

struct sr_prof_rec { int id; hipEvent_t e0, e1; };
struct sr_prof {
    int enabled = 0;
    double ms[SR_K_COUNT] = {0};
    long launches[SR_K_COUNT] = {0};
    std::vector<hipEvent_t> pool;        // recycled events
    std::vector<sr_prof_rec> pending;    // recorded, not yet resolved
    hipEvent_t take() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    // wait for every recorded pair and fold the elapsed times into ms[] / launches[]
    void resolve() {
        for (auto& r : pending) {
            float t = 0.f;
            if (r.e0 && r.e1 && hipEventSynchronize(r.e1) == hipSuccess &&
                hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) {
                ms[r.id] += t;
                launches[r.id] += 1;
            }
            if (r.e0) pool.push_back(r.e0);
            if (r.e1) pool.push_back(r.e1);
        }
        pending.clear();
    }
    void destroy() {
        resolve();
        for (auto e : pool) (void)hipEventDestroy(e);
        pool.clear();
    }
};

// Optional per-launch timing: a hipEvent pair recorded on the launch stream around one kernel.
// Nothing synchronises at launch time; pairs are resolved lazily by sr_prof_get / sr_prof_reset.
struct sr_prof_scope {
    sr_prof* p; sr_prof_rec r;
    sr_prof_scope(sr_prof* p_, int id_, hipStream_t s_) : p(p_), s(s_) {
        r.id = id_; r.e0 = r.e1 = nullptr;
        if (p && p->enabled) {
            if (p->pending.size() >= 8192) p->resolve();
            r.e0 = p->take();
            if (r.e0) (void)hipEventRecord(r.e0, s);
        }
    }
    ~sr_prof_scope() {
        if (p && p->enabled) {
            r.e1 = p->take();
            if (r.e1) (void)hipEventRecord(r.e1, s);
            p->pending.push_back(r);
        }
    }
    hipStream_t s;
};

// ---- launchers implemented in the .hip files ---------------------------------------------------
// C[m][n] (op)= alpha * sum_k A[k][m] * B[k][n]; A: K x M (lda), B: K x N (ldb), C: M x N (ldc).
// M, N multiples of 128; K multiple of 16.
//   mode 0: all tiles.  mode 1: only tiles with n0 >= m0 (upper block triangle).
//   mode 2: B block-lower-triangular (B[k][n] == 0 for k < n0): per tile k starts at n0.
//   mode 4: A block-lower-triangular (A[k][m] == 0 for k < m0): per tile k starts at m0.
//   mode 3: A block-upper-triangular (A[k][m] == 0 for k >= m0 + 128): per tile k ends at m0 + 128.
// prio != 0: the workgroups raise their wavefront priority (critical-path products that share CUs with bulk work)
// Batch of independent, equally shaped problems in ONE launch (the outputs of a model: the same product on n operand
// sets `stride` doubles apart): every launcher below takes an optional sr_batch; n = 1 / NULL = the plain call.
struct sr_batch { int n = 1; long sA = 0, sB = 0, sC = 0, sCT = 0; };
int sr_launch_gemm_tn(const double* A, long lda, const double* B, long ldb, double* C, long ldc,
                      int M, int N, int K, double alpha, double beta, int mode, hipStream_t s, int prio = 0,
                      const sr_batch* bt = nullptr);

// upper block triangle only (tiles n0 >= m0; M <= N) on a linear grid -- the trailing updates of the Cholesky
int sr_launch_gemm_tn_upper(const double* A, long lda, const double* B, long ldb, double* C, long ldc,
                            int M, int N, int K, double alpha, double beta, hipStream_t s, int prio = 0,
                            int order = -1,    // order: 0 row-major tiles, 1 XCD-aware super-tiles, -1 by size
                            const sr_batch* bt = nullptr);
// thin products (N a few tiles, K long): K-slices of ks rows as grid.z into `part` (ceil(K / ks) x M x N doubles), then
// summed in order into C (M x N contiguous).  modes as sr_launch_gemm_tn.
int sr_launch_gemm_tn_splitk(const double* A, long lda, const double* B, long ldb, double* C, int M, int N, int K,
                             int ks, double alpha, int mode, double* part, hipStream_t s);
// a list of independent TN products in one launch (one level of the recursive triangular inversion):
// C_j = alpha A_j^T B_j, optionally also CT_j = C_j^T; operands at offsets (doubles) of common base pointers,
// common leading dimension.  mode 2 / 3 as above.
struct sr_gemm_job { long a, b, c, ct; int M, N, K, pad; };
// tiles128: 128 x 128 tiles of the whole list (picks the workgroup tile)
int sr_launch_gemm_tn_jobs(const double* Ab, const double* Bb, double* Cb, double* CTb, long ld,
                           const sr_gemm_job* jobs_dev, int njobs, int maxM, int maxN, long tiles128, double alpha,
                           int mode, hipStream_t s, const sr_batch* bt = nullptr);

// padded index space: the Np - N padding rows/cols sit at the FRONT (identity), training point i lives
// at padded index i + (Np - N); the contraction kernels simply start at k = 16*floor((Np-N)/16).
// sf2_dev / noise_dev: device scalars that take precedence over the by-value arguments when not NULL
// nbatch > 1: outputs b = 0 .. nbatch-1 in one launch (ls + b D, sf2_dev + b, noise_dev + b, K + b strideK)
int sr_launch_gram(const double* Z, const double* ls, double sf2, double noise, const double* sf2_dev,
                   const double* noise_dev, double* K, int N, int Np, int D, hipStream_t s, int nbatch = 1,
                   long strideK = 0);
// factor the diagonal block kb of the Np x Np matrix A (upper), write U_kk in place, U_kk^-1 to
// wt_diag (into Wt's diagonal block) and U_kk^-T to w_diag (into W's diagonal block).
int sr_launch_append_move(const double* Wt0, int Np0, int off0, int N0, const double* U12t, const double* invS, int m,
                          double* Xt, double* Y2, double* Wt1, int Np1, int off1, hipStream_t s, int nbatch = 1,
                          long sXt = 0, long sY2 = 0);
int sr_launch_eye_front(double* W, int ld, int n, hipStream_t s, int nbatch = 1);
// G != NULL: factor A - G (rows / columns >= pf); nbatch blocks with strides sA (A) and sW (wt_diag), info_dev + b
int sr_launch_potrf_corner16(double* A, long lda, double* wt_diag, long ldw, int* info_dev, hipStream_t s,
                             const double* G = nullptr, int pf = 0, int nbatch = 1, long sA = 0, long sW = 0);
// bt: batch of blocks (sA: stride of A, sB: of wt_diag, sC: of w_diag; info_dev + b)
int sr_launch_potrf_diag(double* A, long lda, double* wt_diag, double* w_diag, long ldw,
                         int kb, int* info_dev, hipStream_t s, int skip = 0, const sr_batch* bt = nullptr);
// one-thread hand-over between streams: publish set_flag = set_v (NULL: nothing), then wait for *w0 >= v0 and *w1 >= v1
// (NULL: no wait); a wait longer than timeout_s sets *status = 1 and passes (sr_factor.hip)
int sr_launch_fact_handover(unsigned* set_flag, unsigned set_v, const unsigned* w0, unsigned v0, const unsigned* w1,
                            unsigned v1, unsigned* status, double timeout_s, hipStream_t s);
int sr_launch_transpose(const double* src, double* dst, int n, hipStream_t s);
int sr_launch_transpose_rect(const double* src, long lds_, double* dst, long ldd, int rows, int cols,
                             hipStream_t s);
// y[r] = sum_c M[r][c] x[c], c in [0..r] (lower=1) or [r..n) (lower=0)
// nbatch members, matrix / x / y sM / sx / sy doubles apart
int sr_launch_trmv(const double* M, long ld, const double* x, double* y, int n, int lower,
                   hipStream_t s, int nbatch = 1, long sM = 0, long sx = 0, long sy = 0);
long sr_mll_ws(int N);
int sr_launch_mll(const double* Kinv, int Np, int N, const double* alpha, const double* yT, const double* Z,
                  const double* kp, int D, const double* logdet, double* partial, double* nll, double* grad,
                  hipStream_t s);
int sr_launch_logdet(const double* Wt, int Np, int n_out, double* out, hipStream_t s);
int sr_launch_fill(double* p, size_t n, double v, hipStream_t s);
int sr_launch_sub_block(double* S, const double* G, int pf, hipStream_t s);
int sr_launch_append_assemble(const double* Wt0, int Np0, int off0, int N0, const double* Y2,
                              const double* invS, int m, double* Wt1, int Np1, int off1, hipStream_t s);

// General kernel family (formulas ssm_gpy/gp_models_utils_casadi.py:17-157):
//   k(x,y) = (c0 + sum_j a_j x_j y_j) * v * kappa(r) + sum_j b_j x_j y_j ,  r^2 = sum_j ((x_j-y_j) s_j)^2
//   kappa = exp(-r^2/2) (RBF, 0) or (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r) (Matern-5/2, 1)
// packed per output as SR_KP(D) doubles: [kappa, v, c0, s[D], a[D], b[D]]
#define SR_KP(D) (3 + 3 * (D))
int sr_launch_gram_general(const double* Z, const double* kp, double noise, const double* noise_dev, double* K, int N,
                           int Np, int D, hipStream_t s, int nbatch = 1, long strideK = 0);

struct sr_kstar_args {
    const double* Z;        // N x D
    const double* alpha;    // n_out x Np
    const double* ls;       // n_out x D
    const double* sf2;      // n_out
    const double* xa; long lda; int na;   // query part a: T x na (row stride lda)
    const double* xb; long ldb; int nb;   // query part b: T x nb (row stride ldb), na+nb == D
    const double* kp;       // general kernels: n_out x SR_KP(D) packed parameters, else NULL (ARD-RBF fast path)
    double* kxx;            // general kernels: prior variance k(x_t, x_t), n_out x Tp
    double* Ks;             // n_out x Np x Tp
    double* mu_part;        // nsplit x n_out x Tp
    double* jac_part;       // nsplit x n_out x D x Tp
    int N, Np, D, n_out, nsplit;
    long T, Tp;
    // one-launch blocking single query (sr_gp_call1; K0 only): the query travels in the kernel arguments, the results
    // go straight to pinned host memory and the last workgroup writes the sequence number the host spins on
    long Tw = 0;            // width (padded queries) of THIS launch when it covers a column range of the buffers only;
                            // 0: all Tp columns.  Pointers (xa, xb, Ks, mu_part, jac_part, kxx) are then pre-offset.
    int xv_on = 0; double xv[SR_MAX_D] = {};
    unsigned* done_ticket = nullptr; unsigned long long* host_flag = nullptr; unsigned long long host_seq = 0;
};
int sr_launch_kstar(const sr_kstar_args& a, hipStream_t s);

// K0 (sr_small.hip): whole posterior of a small ARD-RBF model in one launch, outputs in the API layout
int sr_launch_gp_small(const sr_kstar_args& a, const double* Wt, double* mu, double* var, double* jac,
                       hipStream_t s);
int sr_launch_gp_small_lin(const sr_kstar_args& a, const double* Wt, double* mu, double* var, double* jac_mu,
                           double* jac_var, double* hess_mu, hipStream_t s);

// K0s (sr_server.hip): resident single-query server of a small ARD-RBF model: one workgroup per output polls a mailbox in
// pinned host memory.  All pointers are the DEVICE-visible addresses of pinned host memory.
struct sr_server_args {
    unsigned long long* mb;          // mailbox, ONE 64-byte line: [0 .. 4] x (D <= 5 doubles), [5] (launch epoch << 8) | command (any
                                     // other epoch: leave), [6] sequence number, [7] check word = [0] ^ .. ^ [6] ^ SR_SERVER_CHK
    double* out;                     // reply block: per (output d, part) one record of SR_SERVER_REC doubles
                                     // [mu, var or its share, d mu/dx (D), d var/dx or its shares (D), d2 mu/dx2 (D x D), .., sf2]
    unsigned long long* reply;       // [d]: sequence number last answered by output d; [SR_SERVER_ALIVE + d]: 1 while it runs;
                                     // [2 SR_SERVER_ALIVE + d]: device ticks (100 MHz) of the last evaluation
    unsigned long long first_seq;    // the first sequence number this launch answers
    unsigned long long epoch;        // of this launch (mb[5] holds it while the launch is wanted)
    unsigned long long idle_ticks;   // leave after this long without a request (100 MHz wall clock)
};
#define SR_SERVER_ALIVE 64           /* reply words: one per (output, part) */
#define SR_SERVER_REC 40          /* doubles per output record in the reply block (2 + 2 D + D^2 <= 37 for D <= 5) */
#define SR_SERVER_CHK 0x5afe5eedc0ffee11ull   /* the eight words of a consistent mailbox line xor to this */
#define SR_SERVER_CMD_FIRST 0ull     /* mu, var, d mu/dx */
#define SR_SERVER_CMD_SECOND 1ull    /* + d var/dx, d2 mu/dx2 */
#define SR_SERVER_CMD_STOP 2ull
#define SR_SERVER_CMD_IDLE 3ull      /* (device-internal: the idle time-out) */
#define SR_SERVER_CMD_PING 4ull      /* diagnostics: answer at once, evaluate nothing */
int sr_launch_gp_server(const sr_kstar_args& a, const double* Wt, const sr_server_args& sv, hipStream_t s);

// Persistent multi-step kernel (sr_chain.hip): the whole H-step chain of up to SR_CHAIN_GROUPS / (n_out Np / 128) * 16
// rollouts in ONE launch (the posterior of sr_gp_small_kernel and the step of sr_ellipsoid_kernel inside a loop over the steps).
struct sr_xel { double v; unsigned long long chk; };      // 16 bytes, written and read by single instructions; chk = bits(v) ^ mix(tag)
struct sr_chain_args {
    sr_kstar_args k;                  // model (Z, alpha, ls, sf2, N, Np, D, n_out, na = n_s, nb = n_u); queries unused
    const double* Wt;
    long T; int H; int mode;          // mode as in sr_ell_args
    const double* p0; const double* q0; const double* k_fb0;      // T x n_s, T x n_s^2 | NULL, T x n_u x n_s | NULL
    const double* k_ff; const double* k_fb;                       // T x H x n_u, T x (H-1) x n_u x n_s
    const double* a; const double* b; const double* l_mu; const double* l_sigma; double c_safety;
    double* p_all; double* q_all; double* gp_var_all;             // T x H x n_s, T x H x n_s^2, T x H x n_s | NULL
    int* n_bad;
    sr_xel* xch;                      // groups x H x n_out x (16 (D + 1) + 16 parts) tagged values: every step's results
                                      // of the n_out x Np / 128 posterior workgroups of a group of 16 rollouts (zeroed once)
    // per group: the last tag used so far (the tags of a launch are epoch + 1 .. epoch + H; written by the launch's last
    // workgroup to leave: nothing of the protocol lives on the host, so a captured launch can be replayed), workgroups
    // done, and the tag with which the group's tail workgroup (the shape matrices) reported for this launch
    unsigned long long* epoch; unsigned* done; unsigned long long* alive;
    // status word in PINNED HOST memory (device-visible address): a group whose hand-off did not complete within
    // SR_CHAIN_TIMEOUT_TICKS ORs 1 into it with a system-scope atomic (and poisons its outputs with NaN).  The host
    // reads it without a copy: after the stream has been synchronised it says whether this launch failed.
    int* status;
    int test_drop = 0;                // tests only: launch this many workgroups fewer than the chain needs
};
#define SR_CHAIN_GROUPS 240          /* workgroups of one launch: all must be resident (they wait for each other) */
#define SR_CHAIN_XELS (240ul * 112 * 96)      /* exchange elements (16 B): 240 posterior workgroups x 96 steps at most */
int sr_chain_wgs_per_group(int Np, int n_s);                    // posterior workgroups + the tail workgroup
long sr_chain_xels_per_group(int Np, int n_s, int n_u, int H);  // exchange elements one group of 16 rollouts needs
#define SR_CHAIN_TIMEOUT_TICKS 10000000ull   /* 100 ms of the 100 MHz wall clock (a step takes ~10 us) */
bool sr_chain_supported(int Np, int D, int n_s, int n_u, int H);
// workgroups of the chain kernel for this model that one CU can hold (hipOccupancyMaxActiveBlocksPerMultiprocessor of
// the instantiation the dispatcher would pick; 0 = it cannot run at all); H sizes the dynamic LDS
int sr_chain_blocks_per_cu(int Np, int n_s, int n_u, int H, int* blocks);
int sr_launch_chain(const sr_chain_args& a, hipStream_t s);

// part[d][rb][t] = sum_{i in row block rb} ( sum_k Wt[d][k][i] Ks[d][k][t] )^2
int sr_launch_var(const double* Wt, const double* Ks, double* part, int N, int Np, long Tp, int n_out,
                  int group, int variant, hipStream_t s, long Tw = 0);

// 64 x 64 tile variant for small models (sr_predict.hip, K2m): part layout [d][Np/64][Tp]
int sr_launch_var64(const double* Wt, const double* Ks, double* part, int N, int Np, long Tp, int n_out,
                    hipStream_t s);

// few query tiles: balanced shares of the k-blocks + a reduce pass (sr_predict.hip, K2b); workspace doubles; part layout
// [d][4 nrb][Tp]
long sr_var_bal_ws(int Np, long Tp, int n_out);
int sr_launch_var_bal(const double* Wt, const double* Ks, double* Vt, double* part, int N, int Np, long Tp, int n_out,
                      hipStream_t s);

// small-batch (T <= 16) variance path: U^-1 streamed once at HBM rate (sr_predict.hip, K2s)
#define SR_SMALL_T 16
#define SR_FUSED_T 1024        /* up to here a model with Np <= SR_FUSED_NP takes the one-launch pass of sr_small.hip */
#define SR_FUSED_NP 512
#ifndef SR_STREAM_MIN_NP
#define SR_STREAM_MIN_NP 384    /* T <= 16 streams U^-1 (K2s) above this padded size unless the one-launch pass K0 takes it */
#endif
#define SR_STREAM_MAX_T 64     /* up to here a batch beyond the one-launch sizes streams U^-1 once (sr_stream.hip) */
#define SR_FINAL_WAVE_T 4096   /* up to here sr_finalize runs one wavefront per (query, output) */

// ================================================================================================================
// DISPATCH TABLE -- every size threshold by which the host side picks a kernel, in one place, with the measurement that
// set it (profiles/<round>_<file>; "r1d" .. "r04" = the round that measured it, MI355X).  Np = padded training points
// (multiple of 128), T = queries of one call, Tp = T padded to 128, D = GP input dimension, nb = Np / 128.
//
//  posterior of ONE call (sr_capi_posterior.hip: gp_pass)
//   K0  one launch (sr_small.hip)            Np <= SR_FUSED_NP (512), T <= SR_FUSED_T (1024), D <= 8; not Np = 512 with
//                                            T <= 128 (streamed: 22-23 against 28 us, r02_latency_grid), not Np = 384 with T = 1
//                                            (SR_ONE_STREAMED_NP: 12.4 against 18.8 us, r02_latency_grid)
//   K2f streamed, fused (sr_stream.hip)      Np > SR_STREAM_MIN_NP (384), T <= SR_STREAM_MAX_T (64): N = 5000 T = 1 54 -> 37 us, T = 64
//                                            143 -> 111 us (r02 / r03_latency_grid); one launch for T <= 4 with D <= 5 unless
//                                            2 <= T <= 4 and Np <= SR_MFMA_SMALL_MAX_NP (2048): N = 700 27 against 21 us, N = 3000 34 / 40
//                                            no K* pass either for 16 columns per workgroup on one-chunk work items, nor for 32 up to
//                                            SR_STREAM_FUSED32_MAX_NCB (8) column blocks: T = 16 N = 2000 / 3000 28 / 39 -> 25 / 34 us (r04_latency_grid)
//   K2s streamed, groups of 16               T <= 16 x sr_var_small_groups_max (300 MB of re-read U^-1): N = 700 T = 128 41 -> 25 us,
//                                            N = 2000 71 -> 40 us, from N = 3000 on the tiles win (r01d_latency_grid)
//   K2b balanced shares (sr_predict.hip)     sr_var_splitk_wanted (nb > 2, <= 1024 plain workgroups): one workgroup per cell below
//                                            256 cells, 256 workgroups up to 8192 cells, 512 beyond (r05_bal_ab); the chunked
//                                            split-K route K2k of rounds 1 - 4 and the XCD slabs K2x of round 4 are gone
//   K2m 64 x 64 tiles                        sr_var64_wanted: Np <= 1024 and < 256 plain workgroups (N = 200: 60 -> 31-37 us, r01d)
//   K2  plain 128 x 128 tiles                everything else (the benchmark regime: 0.89-0.90 of the fp64 MFMA peak, r03_kernel_stats)
//   K3  final stage                          one wavefront per (query, output) up to SR_FINAL_WAVE_T (4096) queries (44 -> 9.8 us at
//                                            N = 5000, T = 1, r01d); K1 splits N down to 16 rows per workgroup there (26.6 -> 8.7 us)
//  single blocking query (host entry points)
//   K0s resident server (sr_capi_server.hip) sr_gp_server_supported: Np <= 512, D <= 5; one 16-wavefront workgroup per output at Np = 128,
//                                            Np / 64 parts of 8 wavefronts from 256 on (U^-1 fragments in registers): __call__ 21.6 -> 9.6 us
//                                            at N <= 128, 27 -> 11.9 at N = 150 (cart-pole), 33 -> 12.8 at N = 350 (r04_call_latency)
//   one command (sr_gp_call1)                where K0 applies (Np <= 384; 512 with second order): 26.4 us at N = 200 (r03_call_latency)
//                                            and on the streamed route from 512 padded rows on (query read from the pinned block,
//                                            results + flag by the final stage's workgroup): N = 1000 32 -> 26 us (r04_call_latency)
//  second order of one query (sr_gp_linearize)  K0 LIN up to Np = 384 (18 us at N = 200), streamed above: one launch for D <= 3
//                                            (SR_LIN_FUSED_MAX_D): N = 5000 61 -> 51 us (r02_latency_grid)
//  multi-step chains (sr_capi_reach.hip)     persistent kernel K0c up to SR_CHAIN_GROUPS workgroups (device CUs - 16 at most),
//                                            1 launch for H <= 2, <= 2 launches for T <= 1024, <= 6 beyond (N = 200 H = 15: 1024
//                                            rollouts 203 against 253 us, 4096 in six launches 608 / 722, r03_chain_bench); one-step
//                                            through it for n_s <= 2 (24.5 -> 21.5 us; n_s >= 3: 25 -> 35 us)
//  model update (sr_capi_update.hip)         panels of sr_fact_panel(nb) blocks (r03_factor_bench and the comment there); streams:
//                                            up to SR_FACT_ONE_STREAM_MAX_NB (15) blocks the caller's stream only (r06_small_update_streams),
//                                            regime 1 up to SR_FACT_CHAIN_MAX_NB (128) blocks -- bulk stream without 32 CUs (64 for
//                                            36 < nb <= 52: N = 5000 5.14 -> 4.99 ms) --, regime 2 beyond (8 CUs for the diagonal
//                                            blocks; a trailing update SR_FACT_FREE_RATIO (2) times longer than the next chain takes
//                                            the whole chip: N = 50000 66.8 -> 67.2 TF, r04_timeline50000); early inversion from
//                                            nb >= 8; GEMM tile: sr_use_tile64* (sr_gemm.hip)
//  row append (sr_capi_append.hip)           one launch for +1 point with Np <= 512 (SR_APPEND1_MAX_NP0, r03_exploration_step), <= 16
//                                            points matrix-vector shaped, 17 .. 128 on the MFMA tile (r03_append_bench); the Python layer
//                                            appends up to N / 5 points and refactorises beyond (break-even r03_growing_model)
// ================================================================================================================
#define SR_ONE_STREAMED_NP 384       /* ONE query at this padded size takes the streamed kernel, not K0 */
#define SR_MFMA_SMALL_MAX_NP 2048    /* 2 .. 4 queries up to here: K1 + MFMA streaming kernel instead of the one-launch VALU kernel */
#define SR_STREAM_FUSED_MAX_D 5      /* one-launch streamed predict (T <= 4) evaluates its own K* columns up to this D */
#define SR_LIN_FUSED_MAX_D 3         /* one-launch streamed linearize up to this D */
#define SR_FACT_CHAIN_MAX_NB 128     /* model update: up to here the chain of diagonal blocks bounds it (stream regime 1) */
#define SR_FLOW_MIN_NB 1000           /* model update: tile-flow Cholesky (sr_flow.hip) from this many blocks ... */
#define SR_FLOW_MAX_NB 0              /* ... up to this many (empty range: opt-in through sr_gp_set_fact_pipeline(h, 3)) */
#define SR_FLOW_BAND 2                /* blocks right of the diagonal block that go in 64 x 64 tiles */
static inline int sr_flow_panel(int nb) { return nb <= 12 ? 2 : (nb <= 28 ? 3 : (nb <= 36 ? 4 : (nb <= 44 ? 6 : 8))); }   /* block rows per panel of the tile flow */
#define SR_FLOW_KEEP_WGS 128          /* tile flow: workgroups that stay to the end ... */
#define SR_FLOW_EXIT_PCT 50           /* ... the others leave after a row task past this percentage of the block rows */
#define SR_FLOW_EXIT_PCT_BIG 65       /* ... beyond 64 block rows (GEMM-bound for longer; N = 14000: 61.0 -> 59.4 ms, profiles/r06_flow.txt) */
#define SR_FLOW_TIMEOUT_S 0.25        /* a wait of the tile flow that lasts longer (+ 1 ps per output and Np^3: sr_capi_update.hip) gives up (the host repeats the update by launches) */
#define SR_FACT_ONE_STREAM_MAX_NB 15 /* model update: up to here every launch stays on the caller's stream (no side streams, no events) */
#define SR_APPEND1_MAX_NP0 512       /* +1 point in ONE launch of one workgroup per output up to this padded size (the grown model: <= 640) */
#define SR_APPEND1G_MAX_NP0 8192     /* +1 point in ONE launch of a grid of workgroups up to this padded size (K* row in LDS) */
#define SR_APPEND1G_MAX_W 128        /* workgroups per output of that grid */
#define SR_SLIDE_STEPS SR_NB         /* in-place one-point appends a set of model buffers can take: zeroed slack behind U^-1 (SR_SLIDE_STEPS (Np + 1) doubles), alpha and yT (SR_SLIDE_STEPS each) */
#define SR_STREAM_FUSED32_MAX_NCB 8   // 32 columns per workgroup are evaluated inside the MFMA kernel up to this many 256-column blocks (Np <= 2048)

static inline bool sr_gp_small_wanted(int Np, long T, int D, bool general) {
    (void)general;   // ARD-RBF and the general family both have a one-launch kernel
    if (Np == 512 && T <= 128) return false;  // measured: streaming U^-1 (K2s, groups of 16 queries) 22-23 us against 28 us here
    return Np % 128 == 0 && Np <= SR_FUSED_NP && T <= SR_FUSED_T && D <= 8;   // (D > 8: the hoisted training rows do not fit 128 VGPRs)
}
// ONE query with second-order outputs in one launch (K0 LIN): the ARD-RBF form up to D = 8, the general family (its own phase A:
// two MFMA products, rows 9 + j of the first) up to D = 5
static inline bool sr_gp_small_lin_wanted(int Np, int D, bool general) {
    return sr_gp_small_wanted(Np, SR_SMALL_T, D, general) && (!general || D <= 5);
}
// workgroups per output of the resident server: from Np = 256 on an output is served in Np / 64 parts of eight wavefronts
// (the U^-1 fragments of a part fit its registers), Np = 128 by one workgroup of sixteen
// (a general-family model -- mat52 / lin_* -- is served in parts at every size: two at Np = 128)
constexpr int sr_gp_server_parts(int Np, bool general) { return (Np >= 256 || general) ? Np / 64 : 1; }
// D <= 5 (pendulum: 3, cart-pole: 4 or 5): the D = 6 .. 8 instantiations of the sixteen-wavefront kernel spill and are not built
static inline bool sr_gp_server_supported(int Np, int D) { return Np % 128 == 0 && Np <= SR_FUSED_NP && D <= 5; }
// 64 x 64 tiles: profitable when the model is small and the 128-tile grid would leave most of the chip idle
static inline bool sr_var64_wanted(int Np, long Tp, int n_out) {
    const long wgs128 = (long)(Np / SR_NB) * (Tp / SR_NB) * n_out;
    return Np <= 1024 && wgs128 < 256;
}
// split-K / balanced shares: profitable when the plain kernel cannot fill the chip and there is a K range to split
static inline bool sr_var_splitk_wanted(int Np, long Tp, int n_out) {
    const int nrb = Np / SR_NB;
    const long wgs = (long)nrb * (Tp / SR_NB) * n_out;
    return nrb > 2 && wgs <= 1024;
}
// workgroups of the balanced launch: one per CU, two from 8192 cells on.  Round 5, with the pipelined main loop (G = 256
// against 512, n_out = 2, profiles/archive/r05_bal_ab.txt): N = 5000 T = 128 (U = 1640) 142 / 148 us, T = 256 (3280) 235 / 250, T = 512
// (6560) 426 / 425, T = 1024 (13120) 801 / 795; N = 4000 T = 512 (4224) 291 / 306, T = 1024 (8448) 525 / 535; N = 3000 T = 1024
// 321 / 329 -- the threshold of round 3 (2304 cells, old loop) sent T = 256 at N = 5000 to 512 workgroups; 384, 768 or
// 1024 workgroups -- shares that do not line up with the residency of the chip -- lose 15 - 30 %.  Half of a 512-way launch's
// time goes to twice the partial products (128 KB each, written and read again by the reduce pass).
static inline long sr_var_bal_wgs(long U) {
    static const long forced = sr_lab_env("SR_BAL_WGS", 0);     // (lab build: scripts/bal_ab.py)
    static const long thr = sr_lab_env("SR_BAL_THR", 8192);
    if (forced > 0) return U < forced ? U : forced;
    return U >= thr ? 512 : (U >= 256 ? 256 : U);
}
// The MFMA streaming kernel also serves 17 .. 1024 queries as groups of 16 (every group re-reads U^-1 from
// L2 / Infinity Cache) as long as that stays cheap: n_out Np^2/2 8 B x groups <= 300 MB.  Measured at N = 700,
// T = 128: 41 -> 25 us against the split-K tiles; N = 2000: 71 -> 40 us; from N = 3000 on the tiles win.
static inline int sr_var_small_groups_max(int Np, int n_out) {
    const double bytes = (double)n_out * Np * (double)Np * 4.0;
    int g = (int)(300e6 / bytes);
    if (g > 64) g = 64;
    return g < 1 ? 1 : g;
}
// blocks per Cholesky panel by the number of 128-blocks (measurements: sr_capi_update.hip, pick_fact_panel)
// (round 5, with the pipelined GEMM loops: at N = 50000 panels of 12 .. 24 blocks 69.0 - 69.2 TF, 32: 68.5, 48 -- round 4's choice --
//  67.8, 64: 66.5 on one box, profiles/archive/r05_c4_panels.txt; N = 10000 / 20000 / 30000 keep 8 / 8 / 24)
static inline int sr_fact_panel(int nb) { return nb <= 28 ? 2 : (nb <= 44 ? 3 : (nb <= 64 ? 4 : (nb <= 200 ? 8 : 24))); }
// CUs the bulk streams of the model update leave to the critical chain (regime 1: one per shader engine, two for 36 < nb
// <= 52; regime 2: one per XCD for the diagonal blocks)
// regime 2 of the model update (nb > SR_FACT_CHAIN_MAX_NB): a trailing update this many times longer than the next panel's
// chain runs on the unmasked bulk stream (whole chip; the chain's diagonal blocks then wait for their CUs, hidden)
#define SR_FACT_FREE_RATIO 2.0
static inline int sr_fact_reserved_cus(int regime, int nb) { return regime == 2 ? 8 : ((nb > 36 && nb <= 52) ? 64 : 32); }
// launches of the persistent chain kernel a batch of T rollouts over H steps may take before the per-step route is better
static inline long sr_chain_max_launches(long T, int H) { return H <= 2 ? 1 : (T > SR_FUSED_T ? 6 : 2); }
long sr_var_small_ws(int Np, int n_out);
int sr_launch_var_small(const double* Wt, const double* Ks, double* Vp, double* part, int N, int Np,
                        long Tp, int n_out, int T, hipStream_t s, int dot0 = 0, bool reduce = true);

int sr_launch_var_small_gather(const double* Vp, double* v, int Np, int n_out, int t, int T, hipStream_t s);
int sr_launch_var_small_gather_all(const double* Vp, double* v, int Np, int n_out, int T, hipStream_t s);
int sr_launch_append_alpha(const double* alpha0, int Np0, int N0, const double* Y2, const double* invS,
                           const double* mu_part, int nsplit, int n_out, int d, long Tp, const double* Ynew, int m,
                           double* alpha1, int Np1, hipStream_t s, int qoff = 0, int nbatch = 1, long sY2 = 0);
// one new point on a small model (Np0 <= 512, Np1 <= 640; kp != NULL: general kernels), all outputs, ONE launch (sr_append.hip);
// logdet: n_out x SR_APPEND1_WGS partial sums
#define SR_APPEND1_WGS 8
#define SR_APPEND1_MAX_OUT 16   // outputs whose new targets fit the kernel arguments (sr_gp_append1_host)
long sr_append1_grid_ws(int Np0, int n_out);
int sr_launch_append1_grid(const double* Wt0, const double* alpha0, const double* yT0, const double* Z, const double* ls,
                           const double* sf2, const double* noise, const double* kp, const double* znew, const double* ynew, double* Wt1,
                           double* alpha1, double* yT1, double* Zdst, double* logdet, int* info, int N0, int Np0, int Np1,
                           int D, int n_out, int W, double* ws, unsigned* cnt, unsigned base, unsigned q0, hipStream_t s,
                           const double* x_host = nullptr, const double* y_host = nullptr, int inplace = 0);
#define SR_APPG_ABORTED (-2)         /* status word of every output when the grid of sr_append1_grid_kernel did not assemble */
int sr_launch_append1_small(const double* Wt0, const double* alpha0, const double* yT0, const double* Z, const double* ls,
                            const double* sf2, const double* noise, const double* kp, const double* znew, const double* ynew, double* Wt1,
                            double* alpha1, double* yT1, double* Zdst, double* logdet, int* info, int N0, int Np0, int Np1,
                            int D, int n_out, hipStream_t s, const double* x_host = nullptr, const double* y_host = nullptr);
int sr_launch_append_small(const double* U12t, const double* Wt0, int Np0, int m, int stage, double* G,
                           const double* invS, double* Xt, double* Y2, hipStream_t s, int nbatch = 1);

struct sr_final_args {
    const double* mu_part; const double* jac_part; const double* var_part; const double* sf2;
    const double* ls; const double* kxx;    // kxx != NULL: per-query prior variance instead of sf2
    double* mu; double* var; double* jac;   // T x n_out, T x n_out, T x n_out x D (jac may be NULL)
    int n_out, D, nsplit, nrb; long T, Tp;
};
int sr_launch_finalize(const sr_final_args& a, hipStream_t s);

struct sr_ell_args {
    long T; int n_s, n_u;
    const double* p; long ldp;        // T x n_s, row stride ldp
    const double* q; long ldq;        // T x n_s*n_s or NULL
    const double* k_ff; long ldkff;   // T x n_u
    const double* k_fb; long ldkfb;   // T x n_u*n_s or NULL
    const double* mu; const double* var; const double* jac;   // dense T x n_s, T x n_s, T x n_s x D
    const double* a; const double* b; const double* l_mu; const double* l_sigma;
    double c_safety;
    double* p_out; long ldpo;         // T x n_s
    double* q_out; long ldqo;         // T x n_s*n_s
    int* n_bad;                       // device counter (nullable): queries whose box bounds were not > 0
    int mode;                         // 0 robust ellipsoid (gp_reachability.py), 1 Taylor / 2 mean-equivalent
                                      // Gaussian moment propagation (uncertainty_propagation_casadi.py)
};
int sr_launch_ellipsoid(const sr_ell_args& a, hipStream_t s);
int sr_launch_remainder(long T, int n_s, int n_u, const double* q, const double* k_fb,
                        const double* l_mu, const double* l_sigma, double* u_mu, double* u_sigma,
                        hipStream_t s);
int sr_launch_sample(long T, int size, int n_out, int n_u, const double* mu, const double* var,
                     const double* eps, double* S, const double* k_fb, const double* k_ff, double* z,
                     hipStream_t s);
int sr_launch_distance(long T, int K, int n_s, const double* samples, int per_t, const double* p,
                       const double* q, double* d, hipStream_t s);
int sr_launch_safety(long T, int n_s, int m, const double* p, const double* q, const double* h_mat,
                     const double* h_vec, double c, double* d, hipStream_t s);

// ---- fused small-batch posterior (sr_stream.hip) ------------------------------------------------------
struct sr_lin_args;
// ---- single-query second-order outputs (sr_linearize.hip) --------------------------------------
struct sr_lin_args {
    const double* Z; const double* alpha; const double* ls; const double* Ks; const double* g;
    const double* sf2;               // ARD-RBF signal variances (n_out)
    const double* x;                 // D query coordinates (or the first na of them when xb is set)
    const double* xb = nullptr; int na = 0;   // streamed route: coordinates na .. D-1 come from xb
    const double* kp;                // general kernels: n_out x SR_KP(D) packed parameters, else NULL (ARD-RBF)
    double* jac_var; double* hess_mu;   // n_out x D, n_out x D x D
    int N, Np, D, n_out; long Tp;
};
int sr_launch_trmv_t(const double* M, long ld, const double* x, long xs, double* y, int n, hipStream_t s);

// fused small-batch posterior (sr_stream.hip): up to 128 columns through U^-1 in one pass, reduction and final stage
// behind tickets.  mode 0: columns = queries (predict), fa describes the final stage; mode 1: columns = [k*, dk*/dx]
// of one query (linearize), la / lin_part / nblk / lin_dt / lmu.. describe it.
struct sr_stream_args {
    const double* Wt; const double* Ks; double* Vp; double* part; unsigned* tickets;
    int N, Np, D, n_out, ncb, npairs, k_lo, ncols, ncols_pad, dot0, mode;
    int width_min;                   // take at least this kernel width (16: the MFMA kernel also for <= 4 columns)
    long Tp;
    // columns evaluated in the kernel (ARD-RBF): model and queries
    const double* Z; const double* alpha; const double* ls; const double* sf2;
    const double* xa; long lda; int na; const double* xb; long ldb;
    double* mu_part_w; double* jac_part_w; double* lin_part_w;
    sr_final_args fa;
    sr_lin_args la; const double* lin_part; int nblk, lin_dt; double* lmu; double* lvar; double* ljac_mu;
    // one-command single query (sr_gp_call1 on a streamed model): the workgroup that runs the final stage publishes this
    // sequence number into pinned host memory after its outputs (which then live in pinned host memory too)
    unsigned long long* host_flag; unsigned long long host_seq;
    int probe = 0;                   // measurements only (SR_ST1_PROBE): 1 = sr_stream1_kernel stops after its streaming part
    // polling finaliser of the one-column kernel (round 6): dedicated, self-validating slots (SR_ST1_EMPTY until written):
    // [d][cb] column-block norms, then [j][d][1 + D] the N-split partial sums of mean and mean-Jacobian; NULL = tickets
    double* slots = nullptr;
    // work items of the run kernel (sr_stream_items): device table [column block, run, slot] x nitems, rows per run
    // followed by the workgroups' first entries (nwg + 1): a workgroup runs entries [first[w], first[w + 1])
    const int* item_tab = nullptr; int kr = 0, nitems = 0, nwg = 0;
    int epi = 1;                     // partial products of the MFMA kernels: 1 one 32-byte run per lane; lab build: 0 8-byte pieces (round 5), 2 non-temporal
};
#define SR_ST1_SLOTS_MAX 1536        /* doubles of LDS the finaliser gathers the slots in */
__host__ __device__ static inline long sr_st1_slots(int ncb, int n_out, int D) { return (long)n_out * ncb * (1 + 2 * (1 + D)); }
// bit pattern no arithmetic produces (a signalling NaN with a payload): "not written yet"
#define SR_ST1_EMPTY 0x7FF4C0FFEE5AFE01ull
long sr_stream_vp_doubles(int Np, int n_out, int ncols);
int sr_stream_tickets(int Np, int n_out);
int sr_stream_width(int ncols);
// can_fuse: the caller can have the kernel evaluate its own K* rows (ARD-RBF, D <= 5): one-chunk items then reach further
void sr_stream_plan(int Np, int n_out, int nc, int* g, int* kc, bool can_fuse = false);
#ifdef __cplusplus
#include <vector>
int sr_stream_items(int Np, int n_out, int nc, int n_cu, bool can_fuse, std::vector<int>& tab, int* nitems, int* nwg);   // rows per run (0: no run kernel)
#endif
int sr_launch_stream(sr_stream_args a, int src, hipStream_t s);
int sr_launch_linearize(const sr_lin_args& a, hipStream_t s);
int sr_lin_nacc(int D);
int sr_launch_lin_columns(const sr_lin_args& a, int tq, double* Ks, double* lin_part, hipStream_t s);
int sr_launch_lin_final(const sr_lin_args& a, const double* lin_part, const double* dots, int ncb, double* mu,
                        double* var, double* jac_mu, hipStream_t s);
