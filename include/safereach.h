/* safereach.h -- C-ABI of libsafereach.so (MI355X / gfx950).
 *
 * Drop-in boundary for the GP-dynamics inference + ellipsoid reachability hot path of
 * befelix/safe-exploration.  The reference has NO native interface (pure Python); the Python
 * surface it exposes is mirrored by safe_exploration_amd/ and every entry point below names the
 * reference function(s) it replaces (paths relative to /root/reference/safe_exploration/).
 *
 * Conventions
 *   - plain C, fp64, row-major; every array argument is a DEVICE pointer owned by the caller
 *     unless marked [host]; the library owns only what hangs off its opaque handle.
 *   - every function returns 0 on success, <0 on error (SR_E*); sr_last_error() returns a
 *     thread-local message.  Nothing throws across the boundary.
 *   - work is enqueued asynchronously on the hipStream_t passed as `void* stream` (NULL = default
 *     stream).  A handle is bound to one device and is not thread-safe.
 *   - n_out (GP outputs) == n_s for the reachability entry points; D = n_s + n_u (D = n_x_in + n_u with an input
 *     transform, sr_gp_set_input_transform).
 */
#ifndef SAFEREACH_H
#define SAFEREACH_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sr_gp* sr_gp_t;

#define SR_OK          0
#define SR_EINVAL     -1   /* bad argument / shape                        */
#define SR_EHIP       -2   /* HIP runtime error (no device, OOM, launch)  */
#define SR_ENOTPD     -3   /* Cholesky breakdown: matrix not positive definite */
#define SR_ESTATE     -4   /* call order violated (e.g. predict before factorize) */
#define SR_EBUSY      -6   /* transient: the device could not take the request now (a grid of co-resident workgroups did
                            * not assemble).  Nothing was changed; retry, or take the entry point the comment names. */
#define SR_EUNSUPPORTED -5 /* dimension outside compiled range (n_s<=8, n_u<=4, D<=12).  The reference's systems (n_s <= 4,
                            * D <= 5) run in registers, and so does everything up to n_s = 8 with n_u <= 2; the ellipsoid
                            * step with n_s = 8 and n_u = 3, 4 is compiled but spills 164 / 452 B per lane to scratch
                            * (profiles/archive/r03_kernel_resources.txt): correct, tested, and several times slower per query
                            * than the small instantiations */

/* kernel ids for sr_prof_get() */
#define SR_K_GRAM      0
#define SR_K_POTRF     1
#define SR_K_GEMM      2
#define SR_K_KSTAR     3   /* RBF cross-covariance + mean + mean-Jacobian              */
#define SR_K_VAR       4   /* triangular fp64-MFMA contraction |W k*|^2 (dominant)     */
#define SR_K_FINAL     5
#define SR_K_ELL       6   /* ellipsoid propagate/sum                                  */
#define SR_K_TRINV     7   /* GEMMs of the blocked triangular inversion W = U^-T            */
#define SR_K_SMALL     8   /* one-launch posterior of a small model (Np <= 512, T <= 1024)  */
#define SR_K_COUNT     9

int         sr_version(void);
const char* sr_last_error(void);
/* number of visible HIP devices; SR_EHIP (and *n = 0) if the runtime reports none. */
int         sr_device_count(int* n);

/* ---- model life cycle --------------------------------------------------------------------
 * replaces: SimpleGPModel.__init__/train(opt_hyp=False)/update_model  ssm_gpy/gaussian_process.py:32-70,189-278,347-419
 * One handle = n_out independent ARD-RBF GPs over the same N training inputs Z (N x D). */
int sr_gp_create (sr_gp_t* h, int device, int N, int D, int n_out);
int sr_gp_destroy(sr_gp_t h);

/* Z N x D, Y N x n_out, lengthscale n_out x D, signal_var n_out, noise_var n_out (noise_var must
 * already include every diagonal term: sigma_n^2 + noise_diag(1e-5) + GPy's 1e-8 jitter). */
int sr_gp_set_data(sr_gp_t h, const double* Z, const double* Y, const double* lengthscale,
                   const double* signal_var, const double* noise_var, void* stream);

/* Same, for the general kernel family of the reference's non-RBF kernel types
 * (ssm_gpy/gp_models_utils_casadi.py:43-157: mat52, lin_rbf, lin_mat52):
 *   k(x,y) = (c0 + sum_j a_j x_j y_j) * v * kappa(r) + sum_j b_j x_j y_j,  r^2 = sum_j ((x_j - y_j) s_j)^2
 *   kappa(r) = exp(-r^2/2) (0) | (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r) (1)
 * kparams [device]: n_out x (3 + 3 D) doubles, per output [kappa, v, c0, s_1..s_D, a_1..a_D, b_1..b_D]
 * (s_j = 1/lengthscale_j on the dimensions the stationary factor acts on, 0 elsewhere). */
int sr_gp_set_data_general(sr_gp_t h, const double* Z, const double* Y, const double* kparams,
                           const double* noise_var, void* stream);

/* K_d = rbf(Z,Z) + noise I ; K_d = U^T U (blocked fp64-MFMA Cholesky) ; W = U^-T ; alpha = K^-1 y.
 * replaces the GPy posterior the reference caches as inv_K/beta (ssm_gpy/gaussian_process.py:255-275).
 * info [host, n_out ints]: 0, or 1-based index of the first non-positive pivot (then returns SR_ENOTPD).
 * Synchronises the stream before returning. */
int sr_gp_factorize(sr_gp_t h, void* stream, int* info);

/* Condition on m <= 128 ADDITIONAL training points (same hyper-parameters) without refactorising: block row append of
 * the triangular factor, O(N^2 m).  Znew m x D, Ynew m x n_out; info [host, n_out] like sr_gp_factorize.
 * replaces: update_model(x, y, opt_hyp=False, replace_old=False)  ssm_gpy/gaussian_process.py:347-419
 * (the reference refactorises; its own row-append sketch is ssm_pytorch/utilities.py:74-117).
 * On success N grows by m and Np may grow; no buffer of the factor's size is allocated while Np stays.  m <= 16 also leaves
 * log det of the grown model on the host (sr_gp_logdet_cached).  Blocks the host.  Routes by m: docs/HISTORY.md. */
int sr_gp_append(sr_gp_t h, const double* Znew, const double* Ynew, int m, void* stream, int* info);
/* ONE new point given in HOST memory (x_host: D doubles, y_host: n_out doubles; any host memory): the reference's
 * exploration loop after every step (exploration_runner.py:186-188 -> update_model(x, y, replace_old=False)).
 * The point travels in the kernel arguments, status and log det come back through a pinned block: no copy command.
 * Only where a one-launch append applies (<= 8192 padded rows, n_out <= 16, small path on).
 * SR_EUNSUPPORTED otherwise and SR_EBUSY when the grid of co-resident workgroups could not assemble this time -- either
 * way nothing was touched: copy the point to the device and call sr_gp_append.
 * Beyond 512 padded rows, while Np stays, the point is appended IN PLACE (the model's buffers become views one step further
 * into their allocations; calls that rewrite the model, and big batches after an odd number of steps, copy it back first). */
int sr_gp_append1_host(sr_gp_t h, const double* x_host, const double* y_host, void* stream, int* info);

/* padded leading dimension Np (multiple of 128) of the factor matrices.  The Np - N padding rows and
 * columns are at the FRONT (identity block): training point i has padded index i + (Np - N). */
int sr_gp_padded_n(sr_gp_t h, long* Np);

/* current shape of the model: training points, input dimension, outputs, padded size (any pointer may be NULL) */
int sr_gp_dims(sr_gp_t h, int* N, int* D, int* n_out, long* Np);

/* Export / import the cached posterior state (what a broadcast receiver needs):
 * alpha n_out x N ; Wt n_out x Np x Np = U^-1 (upper triangular, row-major, zero below the diagonal).
 * Either pointer may be NULL.  import marks the handle as factorized. */
int sr_gp_export(sr_gp_t h, double* alpha, double* Wt, void* stream);
int sr_gp_import(sr_gp_t h, const double* alpha, const double* Wt, void* stream);

/* The same state PACKED for the one-time replication to the other GPUs (the reference has no multi-GPU
 * path, so nothing of it is replaced): of U^-1 only the N (N + 1) / 2 entries on and above the diagonal of the real
 * rows travel -- 100 MB instead of 210 MB per output at N = 5000.  Training row i (0 <= i < N) contributes
 * U^-1[i][i .. N-1]; the rows [row0, row1) of output d are packed back to back into buf (sr_gp_packed_count doubles;
 * -1 for a bad range), so a sender can move the factor in bounded pieces through a staging buffer of any size.
 * Receiver: sr_gp_import_begin(alpha n_out x N) after sr_gp_set_data, sr_gp_import_packed for every piece of every
 * output (any order), sr_gp_import_end -> factorized.  Predictions are bit-identical to the sender's. */
long sr_gp_packed_count(sr_gp_t h, long row0, long row1);
int sr_gp_export_packed(sr_gp_t h, int d, long row0, long row1, double* buf, void* stream);
int sr_gp_import_begin(sr_gp_t h, const double* alpha, void* stream);
int sr_gp_import_packed(sr_gp_t h, int d, long row0, long row1, const double* buf, void* stream);
int sr_gp_import_end(sr_gp_t h);

/* explicit (K + noise I)^-1 of output d, N x N -- the reference's `inv_K[d]` attribute
 * (ssm_gpy/gaussian_process.py:258-262).  Cold path; needs factorize() on this handle. */
int sr_gp_inv_k(sr_gp_t h, int d, double* inv_k, void* stream);

/* ---- batched GP posterior ------------------------------------------------------------------
 * replaces: SimpleGPModel.predict / predictive_gradients  ssm_gpy/gaussian_process.py:546-596
 * (formulas: ssm_gpy/gp_models_utils_casadi.py:17-40,160-197).
 * Xq T x D -> mu T x n_out, var T x n_out (clipped at 1e-15), jac T x n_out x D (may be NULL). */
int sr_gp_predict(sr_gp_t h, const double* Xq, long T, double* mu, double* var, double* jac,
                  void* stream);

/* ---- single query with second-order outputs (the CasADi Jacobian callback) ------------------------
 * replaces: linearize_predict(states 1xn, actions 1xm, jacobians=True)  state_space_models.py:106-138,
 *           consumed at :402-415; reference implementation ssm_pytorch/gaussian_process.py:333-385.
 * x D -> mu n_out, var n_out, jac_mu n_out x D, jac_var n_out x D (d var/dx), hess_mu n_out x D x D.
 * All kernel identifiers (rbf, mat52, lin_rbf, lin_mat52) in closed form. */
int sr_gp_linearize(sr_gp_t h, const double* x, double* mu, double* var, double* jac_mu,
                    double* jac_var, double* hess_mu, void* stream);

/* ---- GP input transform ---------------------------------------------------------------------------
 * replaces: the t_z_gp / a_gp_inp_x arguments of gp_reachability_casadi.onestep_reachability (:60-61,85,94-97) and
 * uncertainty_propagation_casadi.one_step_taylor (:40-47,60): the GP is evaluated at x_gp = Tz x (the journal's cart-pole
 * model drops the cart position: D = 4), its state Jacobian is jac[:, :n_x_in] Tz.  Tz [device] n_x_in x n_out (copied);
 * NULL restores the identity.  While set, the reachability / moment entry points below use n_s = n_out,
 * n_u = D - n_x_in.  Plain predictions are not affected. */
int sr_gp_set_input_transform(sr_gp_t h, const double* Tz, int n_x_in, void* stream);

/* ---- one-step reachability, batched over T queries ------------------------------------------
 * replaces: gp_reachability.onestep_reachability  gp_reachability.py:19-156  (+ utils.py:108-144,
 *   utils_ellipsoid.py:63-94,197-233)
 * p T x n_s ; q T x n_s x n_s or NULL (point branch) ; k_ff T x n_u ; k_fb T x n_u x n_s (required iff q != NULL) ;
 * a n_s x n_s ; b n_s x n_u ; l_mu, l_sigma n_s.  -> p_out T x n_s ; q_out T x n_s x n_s ; var_out T x n_s or NULL.
 * n_bad: device int or NULL, atomically incremented once per query whose box half-widths were not all > 0 -- where the
 * reference raises AssertionError (utils_ellipsoid.py:226-228).  The caller zeroes it. */
int sr_onestep_reach(sr_gp_t h, long T, const double* p, const double* q, const double* k_ff,
                     const double* k_fb, const double* a, const double* b, const double* l_mu,
                     const double* l_sigma, double c_safety, double* p_out, double* q_out,
                     double* var_out, int* n_bad, void* stream);

/* ---- multi-step reachability, batched over T trajectories ------------------------------------
 * replaces: gp_reachability.multistep_reachability  gp_reachability.py:159-212
 * p0 T x n_s ; q0 T x n_s x n_s or NULL ; k_fb0 T x n_u x n_s or NULL (required iff q0 != NULL) ;
 * k_ff T x H x n_u ; k_fb T x (H-1) x n_u x n_s (may be NULL when H == 1).
 * -> p_all T x H x n_s ; q_all T x H x n_s x n_s. */
int sr_multistep_reach(sr_gp_t h, long T, int H, const double* p0, const double* q0,
                       const double* k_fb0, const double* k_ff, const double* k_fb, const double* a,
                       const double* b, const double* l_mu, const double* l_sigma, double c_safety,
                       double* p_all, double* q_all, int* n_bad, void* stream);

/* ---- ellipsoid step alone (GP outputs supplied by the caller: any StateSpaceModel) -----------
 * replaces: the algebra of gp_reachability.py:65-156 after `ssm(x, u)` returned.
 * mu T x n_s ; var T x n_s ; jac T x n_s x (n_s+n_u) (ignored in the point branch, may be NULL). */
int sr_ellipsoid_step(int device, long T, int n_s, int n_u, const double* p, const double* q,
                      const double* k_ff, const double* k_fb, const double* mu, const double* var,
                      const double* jac, const double* a, const double* b, const double* l_mu,
                      const double* l_sigma, double c_safety, double* p_out, double* q_out,
                      int* n_bad, void* stream);

/* ---- Gaussian moment propagation (the CautiousMPC baseline), batched -----------------------------
 * replaces: uncertainty_propagation_casadi.one_step_taylor / multi_step_taylor_symbolic (mode 1)
 *           uncertainty_propagation_casadi.one_step_mean_equivalent / mean_equivalent_multistep (mode 2)
 *           uncertainty_propagation_casadi.py:11-283 (numeric evaluation of the symbolic graphs)
 * Step 0 starts from a point (sigma_0 = None is the only case the reference implements, :124,:170);
 * step i >= 1 uses k_fb[i-1].  mu0 T x n_s ; k_ff T x H x n_u ; k_fb T x (H-1) x n_u x n_s
 * -> mu_all T x H x n_s ; sigma_all T x H x n_s x n_s ; gp_var_all T x H x n_s (GP variances, or NULL). */
int sr_multistep_moments(sr_gp_t h, long T, int H, int mode, const double* mu0, const double* k_ff,
                         const double* k_fb, const double* a, const double* b, double* mu_all,
                         double* sigma_all, double* gp_var_all, void* stream);
/* one step with the GP outputs supplied by the caller (sigma_x NULL = point input). */
int sr_moment_step(int device, long T, int n_s, int n_u, int mode, const double* mu_x,
                   const double* sigma_x, const double* k_ff, const double* k_fb, const double* mu_g,
                   const double* var_g, const double* jac_g, const double* a, const double* b,
                   double* mu_out, double* sigma_out, void* stream);

/* replaces: utils.compute_remainder_overapproximations  utils.py:108-144 (batched)
 * q T x n_s x n_s, k_fb T x n_u x n_s -> u_mu T x n_s, u_sigma T x n_s */
int sr_remainder_overapprox(int device, long T, int n_s, int n_u, const double* q, const double* k_fb,
                            const double* l_mu, const double* l_sigma, double* u_mu, double* u_sigma,
                            void* stream);

/* replaces: gp_reachability.lin_ellipsoid_safety_distance  gp_reachability.py:215-250 (batched)
 * p T x n_s, q T x n_s x n_s, h_mat m x n_s, h_vec m -> d T x m */
int sr_safety_distance(int device, long T, int n_s, int m, const double* p, const double* q,
                       const double* h_mat, const double* h_vec, double c_safety, double* d,
                       void* stream);

/* replaces: utils_ellipsoid.distance_to_center / sample_inside_ellipsoid  utils_ellipsoid.py:16-60 (batched;
 * the Monte-Carlo verification of sampling_models.py / gp_reachability.py:323-356 evaluates it per step)
 * T ellipsoids (p T x n_s, q T x n_s x n_s) against K samples: samples K x n_s shared by all ellipsoids
 * (per_t = 0) or T x K x n_s (per_t = 1) -> d T x K with d = (s-p)^T Q^-1 (s-p). */
int sr_distance_to_center(int device, long T, int K, int n_s, const double* samples, int per_t,
                          const double* p, const double* q, double* d, void* stream);

/* replaces: the objective that GPy's model.optimize() minimises inside SimpleGPModel.train(opt_hyp=True)
 * ssm_gpy/gaussian_process.py:249-250 (the optimiser itself stays on the host: L-BFGS-B over these values).
 * For the factorised model (data set with sr_gp_set_data_general):
 *   nll[d]  = 1/2 y^T alpha + 1/2 log det K_y + N/2 log 2pi
 *   grad[d] = d nll / d [v, c0, s[D], a[D], b[D], noise]   (SR_KP(D) = 3+3D entries per output, the packed
 *             kernel parameters in their order, then the diagonal noise term). */
int sr_gp_mll(sr_gp_t h, double* nll /* n_out, device */, double* grad /* n_out x (3+3D), device */,
              void* stream);

/* replaces: the determinant inside SimpleGPModel.information_gain  ssm_gpy/gaussian_process.py:621-634
 * (log det(I + K/sigma_n^2) = log det(K + sigma_n^2 I) - N log sigma_n^2).
 * logdet[d] = log det(K_d + noise_d I) of the factorised (or imported) model, from the diagonal of U^-1. */
int sr_gp_logdet(sr_gp_t h, double* logdet /* n_out, device */, void* stream);
/* The same numbers from the HOST copy that sr_gp_append (<= 16 rows) reads back together with its status words -- no
 * launch, no synchronisation: the reference's exploration loop asks for the information gain after every appended
 * transition (exploration_runner.py:186-189).  SR_ESTATE when the last model update left no copy (a factorisation, an
 * imported model, an append of more than 16 rows): call sr_gp_logdet then. */
int sr_gp_logdet_cached(sr_gp_t h, double* logdet_host /* n_out, host */);

/* replaces: SimpleGPModel.sample_from_gp  ssm_gpy/gaussian_process.py:598-619 (marginal posterior samples,
 * GPy posterior_samples_f(full_cov=False)) and the propagation step of
 * MonteCarloSafetyVerification.sample_n_step  sampling_models.py:66-80.
 * mu, var T x n_out (from sr_gp_predict), eps T x size x n_out standard-normal draws supplied by the
 * caller -> S T x size x n_out = mu + sqrt(var) * eps.  If z_next != NULL it also receives the next GP
 * inputs T x size x (n_out + n_u): [S, k_fb S + k_ff] with k_fb n_u x n_out, k_ff n_u. */
int sr_gp_sample(int device, long T, int size, int n_out, int n_u, const double* mu, const double* var,
                 const double* eps, double* S, const double* k_fb, const double* k_ff, double* z_next,
                 void* stream);

/* ---- tuning / measurement -------------------------------------------------------------------- */
/* max queries processed per internal pass (workspace = n_out * Np * chunk * 8 B); default 65536. */
int sr_gp_set_chunk(sr_gp_t h, long chunk);
/* query tiles per scheduling group of the variance kernel (L2/XCD locality knob); default 64. */
int sr_gp_set_var_group(sr_gp_t h, int group);
/* main loop of the variance kernel: 1 = the loop of rounds 1 - 4 (LDS-DMA tiles, workgroup barrier on top of every
 * k-tile), 3 = the pipelined loop of round 5 (barrier under the MFMA stream; same results as 1 bit for bit), 4 = 3 with
 * the structural zeros of the diagonal blocks of U^-1 left out (same numbers summed in another order: equal to 1e-13),
 * 5 = 4 with the row blocks (nrb - 1 - p, p) of a query tile in ONE workgroup (equal work per workgroup; bit for bit 4).
 * A measurement knob; other values are SR_EINVAL. */
int sr_gp_set_var_variant(sr_gp_t h, int variant);
/* blocks of 128 rows per Cholesky panel of sr_gp_factorize (the trailing matrix is read-modify-written once per
 * panel); 0 = chosen by size (default).  Results agree to rounding; a measurement knob. */
int sr_gp_set_fact_panel(sr_gp_t h, int panel);
/* How sr_gp_factorize runs between 3 and 128 blocks of 128 rows.  0 (default) = one chain of launches.  1 / 2 = the pipelined
 * prototypes of round 6 (diagonal blocks on a stream of their own / only the rest of the rows moved; identical numbers,
 * 1.2 - 2.8 times slower: profiles/r06_fact_pipeline.txt).  3 = the tile-flow Cholesky (csrc/sr_flow.hip: ONE resident
 * kernel of tile tasks plus a resident diagonal-block workgroup per output, device counters for dependencies; results to
 * rounding, time within 2 % of the launched form from N = 6000 on, behind below: profiles/r06_flow.txt).  -1 = never the tile flow.
 * sr_gp_fact_pipelined: how the last update of h ran -- 0 chain of launches, 1 pipelined prototype, 4 tile flow. */
int sr_gp_set_fact_pipeline(sr_gp_t h, int on);
int sr_gp_fact_pipelined(sr_gp_t h);
/* diagnostics of the last tile-flow update (SR_ESTATE if it was none): out[0..23] = per kind of task (look-ahead update,
 * bulk update, diagonal tiles, near updates, near solves, far blocks) [count, ticks inside, ticks of those waiting, 0] at
 * 100 MHz; then per output the ticks from the start to each diagonal block.  Returns the words written (cap too small:
 * SR_EINVAL). */
int sr_gp_flow_stats(sr_gp_t h, unsigned* out, int cap);
/* latency paths instead of the plain MFMA tiles: one-launch pass for small models (Np <= 512, T <= 1024),
 * HBM-bound streaming of U^-1 for batches of <= 64 queries, 64 x 64 tiles, balanced shares of the k-blocks under few
 * query tiles.  on = 1 (default) all of them, 2 all but the one-launch pass, 0 none; an A/B measurement knob.  Results
 * agree to rounding. */
int sr_gp_set_small_path(sr_gp_t h, int on);
/* multi-step chains (sr_multistep_reach / sr_multistep_moments) of small ARD-RBF models (Np <= 512, <= 4096 rollouts,
 * the reference's systems n_s <= 4) run all H steps inside ONE persistent launch; on = 0 forces the per-step launches.
 * Default on; results agree to rounding.  sr_gp_last_chain: 1 if the last chain took the persistent kernel. */
int sr_gp_set_chain(sr_gp_t h, int on);
/* The workgroups of the persistent kernel wait for each other; they are launched only when the kernel's occupancy
 * (hipOccupancyMaxActiveBlocksPerMultiprocessor) and the device's CU count say they can all be resident.  Should the
 * hand-off of a group still not complete within 100 ms (the CUs were held by other work for that long), the group fills
 * its outputs with NaN AND raises a status word in pinned host memory.  sr_gp_chain_status: call after synchronising the
 * stream; *timed_out = 1 if a launch since the last query failed that way (the word is cleared, and the handle takes the
 * per-step launches from then on).  A caller that does not ask is told by the next reachability entry point, which
 * returns SR_ESTATE once. */
int sr_gp_chain_status(sr_gp_t h, int* timed_out);
/* The model update keeps its scratch (two Np x Np matrices per output in flight) with the handle while that is at most a
 * third of the device's memory, so that refits allocate nothing (40 GB at N = 50000); the row append keeps a strip and
 * the previous U^-1 buffer.  A host that will only evaluate the model from here on hands them back with this call. */
int sr_gp_release_scratch(sr_gp_t h);
/* Device buffers of >= 1 MB that a handle releases (a model that grew, a handle that was destroyed) stay with the library
 * for the next request of their size class -- the first touch of a fresh allocation, not hipMalloc, is what a refit after
 * a change of N used to pay for (48 against 6 ms at N = 5000); at most an eighth of the device's memory is held, an
 * allocation that fails returns them all before it is reported.  This call hands
 * everything cached back to the driver (all devices). */
int sr_release_cached_memory(void);

/* ---- completion mailbox for a host that blocks on ONE small result --------------------------------
 * (the CasADi / IPOPT callback: CasadiSSMEvaluator.eval, state_space_models.py:271-303, calls the model, waits, and
 * hands NumPy arrays back).  sr_publish enqueues, behind the work already on `stream`, a copy of n doubles from device
 * memory into PINNED host memory followed by a store of `seq` to *flag_host (pinned as well); sr_wait_flag spins on the
 * host until *flag_host == seq (SR_ESTATE after timeout_s).  The result is visible a PCIe write after the producing
 * kernel ends; a D2H copy + hipStreamSynchronize takes ~4 us longer (N = 200: 33.7 -> 29.4 us per blocking call). */
int sr_publish(int device, const double* src_dev, int n, double* dst_host, unsigned long long* flag_host,
               unsigned long long seq, void* stream);
int sr_wait_flag(const unsigned long long* flag_host, unsigned long long seq, double timeout_s);
/* 1 where kernels of `device` may be handed the HOST address of this pinned block as it is (device-visible at the same
 * address, and the device's atomic adds -- the n_bad counters -- reach pinned host memory: probed once per device and
 * process with one tiny launch on a word of the library's own), 0 otherwise: asked once by a host layer that lets the kernels read / write its pinned staging
 * blocks directly (tiny batches: no copy command in either direction; safe_exploration_amd/_buffers.py). */
int sr_host_block_is_device_visible(int device, const void* host_block);
/* hipStreamSynchronize(stream) with `device` current, for a host layer that holds the raw stream handle (the handle 0 --
 * the null stream, PyTorch's default -- names the stream of whichever device is current: hence the device). */
int sr_stream_synchronize(int device, void* stream);
/* One blocking single query as ONE command.  replaces SimpleGPModel.__call__ (ssm_gpy/gaussian_process.py:135-144) and
 * linearize_predict(jacobians=True) as CasadiSSMEvaluator drives them (state_space_models.py:271-303, 384-417).
 * x_host (D doubles, read at call time) travels in the kernel arguments; the results go to the pinned block
 *   out_host = [mu n | var n | jac_mu n x D]   (+ [jac_var n x D | hess_mu n x D x D] with second_order != 0)
 * and the last workgroup stores seq to *flag_host (pinned; wait with sr_wait_flag).
 * From 512 padded rows on the streamed kernels read the query from x_host: it must then be pinned, device-visible too.
 * SR_EUNSUPPORTED with an input transform, with the size dispatch switched off, or where a block is not device-visible:
 * use sr_gp_predict / sr_gp_linearize. */
int sr_gp_call1(sr_gp_t h, const double* x_host, int second_order, double* out_host,
                unsigned long long* flag_host, unsigned long long seq, void* stream);
/* Resident single-query server (the MPC loop's latency path).  replaces the blocking model evaluation inside
 * CasadiSSMEvaluator.eval / JacFun.eval / BackFun.eval (state_space_models.py:278-303, 384-417, 534-562).
 * _start launches workgroups that STAY on the device and poll a mailbox line in pinned host memory; _call posts x_host
 * (D doubles, any host memory; GP input space), waits and copies the answer to out_host (layout of sr_gp_call1).
 * Every kernel identifier, Np <= 512, D <= 5; SR_EUNSUPPORTED otherwise and from _call when no server is armed.
 * The kernel leaves after idle_timeout_s without a request and _call launches it again; model updates take it off the
 * device first (it stays armed).  No answer within timeout_s: SR_ESTATE, the next call starts a fresh launch.
 * Threads, device-wide synchronisation, mailbox protocol: INTEGRATION.md 2.3 and csrc/sr_capi_server.hip. */
int sr_gp_server_start(sr_gp_t h, double idle_timeout_s);
int sr_gp_server_stop(sr_gp_t h);
int sr_gp_server_call(sr_gp_t h, const double* x_host, int second_order, double* out_host, double timeout_s);
int sr_gp_server_state(sr_gp_t h, int* armed, int* resident, long* launches, long* calls);
int sr_gp_last_chain(sr_gp_t h);
/* diagnostic: C(M x N) = alpha * A^T B + beta * C with A (K x M), B (K x N) k-major; M, N multiples
 * of 128, K multiple of 16; mode 0 all tiles, 1 upper block triangle, 2 B block-lower-triangular.
 * Exposed so the fp64-MFMA tile can be tested in isolation. */
int sr_test_gemm_tn(int device, const double* A, long lda, const double* B, long ldb, double* C,
                    long ldc, int M, int N, int K, double alpha, double beta, int mode, void* stream);
/* diagnostic: the trailing-update product of the factorisation alone: C (M x N, only the 128-tiles n0 >= m0) =
 * alpha A^T B + beta C, A (K x M), B (K x N) k-major, M <= N multiples of 128; order 0 row-major tiles, 1 XCD-aware
 * super-tiles, -1 chosen by size. */
int sr_test_gemm_tn_upper(int device, const double* A, long lda, const double* B, long ldb, double* C, long ldc,
                          int M, int N, int K, double alpha, double beta, int order, void* stream);
/* diagnostic: the diagonal-block kernel of the factorisation alone: A (128 x 128 SPD, upper triangle read, leading
 * dimension lda) -> upper Cholesky factor in place, wt = its inverse, w = the inverse transposed (leading dimension
 * ldw); info: device int, 0 or the 1-based first non-positive pivot.  skip: 0 in production; 64 leaves A untouched
 * (back-to-back timing on one input). */
int sr_test_potrf_diag(int device, double* A, long lda, double* wt, double* w, long ldw, int* info, int skip,
                       void* stream);
/* diagnostic: the following persistent multi-step launches of the handle are `drop` workgroups short, so that the last
 * group of rollouts waits for partners that never come: the deterministic way into the time-out path (the tests check
 * that it is reported -- sr_gp_chain_status -- and not silent).  0 restores normal launches. */
int sr_test_chain_drop(sr_gp_t h, int drop);
/* host only: the task plan of the tile-flow Cholesky (csrc/sr_flow.h) for nb block rows -- segs[4 i + 0..3] = first critical
 * task, far task, panel update, position in the one order of block row i (i = 0 .. nb: one behind the last); totals[0..3] =
 * critical tasks, far tasks, panel updates, positions per output. */
int sr_test_flow_plan(int nb, int band, int panel, int* segs, long* totals);
/* n > 0: the next n tile-flow model updates of this process fail ON THE DEVICE (their diagonal-block workgroups never get
 * their go: every wait runs into its time-out, the status word is raised) -- sr_gp_factorize must then repeat the update by
 * launches, and the process' next 16 (then 32, .. 4096) would-be tile flows run by launches too; n = 0: forget that (the tile
 * flow runs again at once). */
int sr_test_flow_fail(int n);
/* diagnostic: the next n launches of the one-launch append of a grid of workgroups (sr_gp_append / sr_gp_append1_host with
 * one point beyond 512 padded rows) wait at their first device-wide barrier for a workgroup that does not exist and give
 * it up after ~5 ms: the deterministic way into the path a grid takes that cannot become resident as a whole (nothing of
 * the model written, the append done by separate launches instead; sr_gp_append1_host answers SR_EBUSY).
 * sr_gp_grid_append_aborts: how often that has happened to h so far. */
int sr_test_grid_append_abort(int n);
int sr_gp_grid_append_aborts(sr_gp_t h, long* n);
/* per-kernel hipEvent timing inside the library (adds an event pair per launch while enabled). */
int sr_prof_enable(sr_gp_t h, int on);
int sr_prof_reset (sr_gp_t h);
int sr_prof_get   (sr_gp_t h, int kernel_id, double* ms_total, long* launches);

/* model replication over RCCL for hosts without PyTorch: include/safereach_comm.h (libsafereach_comm.so) */

#ifdef __cplusplus
}
#endif
#endif /* SAFEREACH_H */
