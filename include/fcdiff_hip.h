/*
 * fcdiff_hip.h -- C ABI of libfcdiff_hip.so, the MI355X (gfx950) implementation of the fcdiff
 * fit path.  Plain C: raw device pointers, sizes, a hipStream_t passed as void*.  No torch, no
 * C++ types.  Every function returns 0 on success, a negative FCD_ERR_* for an argument error
 * detected on the host, or a positive hipError_t.  Nothing throws; nothing synchronises the
 * device or allocates unless its comment says so (fcd_ctx_create / fcd_ctx_destroy / fcd_ctx_reserve do; any
 * sampler or fit entry point does ONLY when it meets a shape larger than fcd_ctx_reserve was told: it then grows
 * the context's scratch once, which is a hipDeviceSynchronize + hipMalloc -- fcd_ctx_stat "n_alloc" counts them).
 * There is no global state: all of it sits behind fcd_ctx.  No entry point reads the environment; the tuning /
 * test knobs take their defaults from it once, in fcd_ctx_create.
 *
 * The reference (andy-sweet/fcdiff) is pure Python/NumPy and has NO native interface; what each
 * entry point replaces is therefore a NumPy method of fcdiff/fit.py, cited per function.  The
 * binding a maintainer would add on the reference side is a ctypes stub.
 *
 * Naming follows the reference: Nreg regions ("N" there), C = Nreg(Nreg-1)/2 edges in
 * lower-triangular row-major order c = n(n-1)/2 + m, n > m (fcdiff/util.py:40-84), H healthy
 * subjects, U patients.  All floating point is IEEE binary64.
 *
 * Array layouts (all C-contiguous, device memory unless marked "host"):
 *   b        (C, H)        correlations of healthy subjects          fcdiff/fit.py:20-21
 *   bt       (C, U)        correlations of patients                  fcdiff/fit.py:22-23
 *   S_B      (C, 3)        sum_h log N(b[c,h]; mu_k, sigma_k): the H-sum of _lp_B_g_F, which is all
 *                          fit.py:171 and :472 ever consume
 *   lM       (C, U, 3, 3)  _lM[c,u,k,l], the reference's layout       fcdiff/fit.py:48-49
 *   lq_F     (C, 1, 3)     _lq_F                                     fcdiff/fit.py:42-43
 *   lq_R     (Nreg, U, 2)  _lq_R                                     fcdiff/fit.py:40-41
 *   hyper    (8,)          {ln gamma_0..2, ln(1-pi), ln pi, 0, 0, 0}: the hyper-parameters every
 *                          conditional reads; lives on the device so an M-step never needs the host
 *   theta    (12,) host    {pi, eta, epsilon, gamma[3], mu[3], sigma[3]}  fcdiff/model.py:31-38
 * Chain state of the Gibbs sampler, G chains padded to GW = ceil(G/64) words of 64 chains:
 *   f_state  (GW, C, 64)   uint8 in {0,1,2}: f_c of chain 64*w + lane at [(w*C + c)*64 + lane]
 *   r_bits   (GW, Nreg, U) uint64: bit `lane` of [(w*Nreg + n)*U + u] is r_{n,u} of chain 64*w+lane
 */
#ifndef FCDIFF_HIP_H
#define FCDIFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FCD_ABI_VERSION 4

#define FCD_OK 0
#define FCD_ERR_ARG (-1)         /* null pointer / non-positive size */
#define FCD_ERR_SHAPE (-2)       /* C is not a triangular number (fit.py:62-65), Nreg < 2, ... */
#define FCD_ERR_UNSUPPORTED (-3) /* shape outside what the kernels are built for (message says which) */
#define FCD_ERR_INDEX (-4)       /* reference edge ids run out of range (Nreg == 2, fit.py:186) */
#define FCD_ERR_DEVICE (-5)      /* a kernel gave up a device-side wait (one-launch r pass); the chain state is unusable */
#define FCD_ERR_COMM (-6)        /* RCCL: library not found, or a call failed (fcd_last_message says which) */
#define FCD_COMM_ID_BYTES 128    /* sizeof(ncclUniqueId) */

/* Edge id used by the region update for an ordered pair (n, m), m != n. */
#define FCD_EDGE_REFERENCE 0 /* nm_to_c(n,m) = n(n-1)/2 + m for EVERY ordered pair, as fit.py:185-186 calls it */
#define FCD_EDGE_SYMMETRIC 1 /* the unordered pair's edge, as doc/methods.rst:646-653 writes it */

typedef struct fcd_ctx fcd_ctx;
typedef void *fcd_stream; /* hipStream_t */

int fcd_abi_version(void);
/* Static string for any return code of this library (hipGetErrorString for positive codes). */
const char *fcd_strerror(int code);
/* Last host-side diagnostic of this ctx (which argument / which limit); never NULL. */
const char *fcd_last_message(const fcd_ctx *ctx);

/* Context on the CURRENT hip device: reduction workspace + device properties.  create/destroy
 * allocate/free device memory (they synchronise); nothing else does, except that a call needing a
 * larger workspace than any before grows it (hipDeviceSynchronize + hipMalloc) -- call fcd_ctx_reserve
 * first to avoid that. */
int fcd_ctx_create(fcd_ctx **out);
int fcd_ctx_destroy(fcd_ctx *ctx);
/* Sizes every scratch buffer of the sweep at this shape (f / r pass workspace, square f copy): after it no
 * sampler entry point allocates or synchronises at shapes up to (Nreg, U, G).  Synchronises when it grows something. */
int fcd_ctx_reserve(fcd_ctx *ctx, int64_t Nreg, int64_t U, int64_t G);
/* Tuning / test knobs (defaults: environment FCD_R_PATH, FCD_R_UB, FCD_R_NOPAD, FCD_R_DSPLIT, FCD_R_COOP, FCD_R_REFILL, FCD_R_TOL, FCD_F_TOL, FCD_F_FORM,
 * FCD_CORR_FORM, read once by fcd_ctx_create; 0 = default everywhere; the two test hooks at the end of the list are NOT read
 * from the environment -- a stray variable must not be able to make a fit give up):
 *   "r_path"    0: blocked r pass in its pipelined one-launch form (marks / sentinels in device memory instead of
 *                  kernel boundaries) wherever every workgroup is resident at once, else one launch per block step;
 *               3: one launch per block step always
 *   "r_ub"      1 / 2 / 4: patients per panel workgroup of the blocked r pass (0: chosen by shape)
 *   "r_nopad"   1: no empty workgroups beside the in-order workgroups
 *   "r_dsplit"  1: ONE in-order workgroup per patient in the pipelined r pass (default: two, 8 chain words each, on two CUs,
 *               where a group has more than 8 chain words and 2 U <= number of CUs)
 *   "r_coop"    1: COOPERATIVE launch of the pipelined r pass -- the runtime itself checks that the grid is resident at once
 *               and the pass falls back to one launch per block step if it refuses (measured +16 us per pass at cfg3, so
 *               not the default; the default relies on the occupancy query, 8 spare slots and bounded polls)
 *   "r_refill"  1: the pipelined r pass's packing launch writes the panel-value sentinels in every sweep (default: only in the
 *               first sweep of a fcd_gibbs_run call -- a completed pass leaves every slot holding its sentinel again)
 *   "qr_form"   fcd_vb_update_qR: 1 = operands gathered from the edge-major table inside the region loop (rounds 1-3; the
 *               default where the region-major weights would exceed 192 MB: cfg5), 2 = region-major weights made first
 *               whatever their size (the default below that: cfg3)
 *   "corr_form" 1: fcd_corr_edges in 64 x 64 blocks with a moments pass also where the one-workgroup-per-subject kernel
 *               (up to 208 regions) would run
 *   "r_tol", "f_tol"  widen the margin inside which a fast r / f draw is repeated with the exact formula (1e30: all)
 *   "f_form"    2: the any-U pair kernel of the f pass also where the U <= 64 kernel would run; 3: scalar-mask form
 *   "r_poll_limit", "r_withhold"  TEST HOOKS of the pipelined r pass: bound every device-side poll by this many polls /
 *               the in-order role never announces a block (a panel wave then gives its wait up, fcd_ctx_check reports it)
 * None of them changes a result: every combination walks the same chains (tests/test_gpu_parity.py).
 * (Round 2 carried eight more forms behind knobs -- a one-launch form with counters, a row-sequential kernel, two-stream
 * half-passes, a pair-record table, triple records in the f pass, ... -- all measured slower; DESIGN.md keeps the numbers.) */
int fcd_ctx_set_knob(fcd_ctx *ctx, const char *name, double value);
/* Counters of the context: "n_alloc" device allocations made so far, "ws_bytes", "fsq_bytes"; "r_form_last" = the form
 * the last blocked r pass ran in (1 one launch per block step, 2 pipelined one-launch form, 3 one-launch form with
 * counters); "dev_err" = the error word of the pipelined r pass as the host sees it now (see fcd_ctx_check). */
int fcd_ctx_stat(const fcd_ctx *ctx, const char *name, int64_t *out);
/* FCD_ERR_DEVICE if a kernel of this context has abandoned a device-side wait (pipelined r pass: every poll of a mark or
 * of a panel value is bounded, ~1 s), else FCD_OK.  The word is written by the device: call this AFTER the stream has been
 * synchronised (or after any device-to-host read that follows the sweeps on that stream) -- the sampler entry points
 * themselves only see a give-up of an EARLIER call.  Once set, the chain state is unusable; fcd_ctx_clear_error resets
 * the word after the caller has re-initialised its chains, and puts the context-owned accumulators and tickets (pooled
 * counts of the tally, K_corr's per-subject tickets) back to zero (it synchronises: an error-recovery call). */
int fcd_ctx_check(fcd_ctx *ctx);
int fcd_ctx_clear_error(fcd_ctx *ctx);

/* ---- the one exchange between GPUs: pooled counts before a (pi, gamma) M-step --------------------------------------
 * ("fcd_allreduce_stats(ncclComm_t, ...)"; the reference has no counterpart -- single process.)
 * One process per GPU.  The communicator belongs to the context and is made ONCE, before any sweep:
 *   fcd_comm_load(ctx, path)     take RCCL's entry points from the librccl the process already holds (path = the file the
 *                                host framework loaded, e.g. <torch>/lib/librccl.so; NULL / "" tries librccl.so.1)
 *   fcd_comm_unique_id(ctx, id)  rank 0: ncclGetUniqueId -> 128 bytes, broadcast to the other ranks by whatever the caller
 *                                has (the Python mirror uses its torch.distributed group)
 *   fcd_comm_init(ctx, id, world, rank)   every rank (collective): ncclCommInitRank
 * From then on fcd_gibbs_run pools the counts of every M-step over the communicator's ranks with ncclAllReduce(8 x int64,
 * sum) ON THE STREAM OF THE SWEEP KERNELS, between the tally and a one-thread M-step kernel: no copy of the counts, no
 * event between streams, no host in the loop.  A communicator of one rank is legal (and tested on the one-GPU box).
 * fcd_allreduce_stats does the same for a caller-owned counts vector (8 int64, device), e.g. after fcd_gibbs_stats.
 * fcd_comm_destroy before fcd_ctx_destroy (which also calls it). */
int fcd_comm_load(fcd_ctx *ctx, const char *librccl_path);
int fcd_comm_unique_id(fcd_ctx *ctx, uint8_t *id128);
int fcd_comm_init(fcd_ctx *ctx, const uint8_t *id128, int world, int rank);
int fcd_comm_destroy(fcd_ctx *ctx);
int fcd_allreduce_stats(fcd_ctx *ctx, int64_t *counts, fcd_stream stream);

/* Optional timing of the library's main kernels with HIP events recorded on the launch stream, each pair
 * bracketing exactly ONE kernel launch.  slot: 0 likelihood tables, 1 f pass, 2 r block step (or one-launch pass), 3 r pack.
 * fcd_prof_collect waits for the recorded events, returns the summed milliseconds and the number of
 * launches, and clears the slot.  Off by default (no events are recorded). */
int fcd_prof_enable(fcd_ctx *ctx, int on);
int fcd_prof_collect(fcd_ctx *ctx, int slot, double *total_ms, int64_t *count);

/* ---- index maps: fcdiff/util.py:7-84 (host, integer) ------------------------------------- */
int64_t fcd_N_to_C(int64_t Nreg);
/* Returns Nreg with C = Nreg(Nreg-1)/2, or FCD_ERR_SHAPE when C is not triangular (fit.py:62-65). */
int64_t fcd_C_to_N(int64_t C);
int64_t fcd_nm_to_c(int64_t n, int64_t m);
int fcd_c_to_nm(int64_t c, int64_t *n, int64_t *m);

/* ---- hyper-parameter block ---------------------------------------------------------------- */
/* Writes hyper[0..7] from host values: gamma[3] and the 2-vector pi2 = {1-pi, pi} the reference
 * indexes (quirk Q4: fit.py:183, :486; test_fit.py:208, 477-487).  Asynchronous on `stream`. */
int fcd_hyper_set(fcd_ctx *ctx, double *hyper, const double *gamma3_host, const double *pi2_host,
                  fcd_stream stream);

/* ---- likelihood tables: UnsharedRegionFit._update_lps, fit.py:104-122 + _eval_M 409-444 ----
 * Same arithmetic as the reference: linear-space Normal densities, M_kl = eps_l N_k +
 * (1-eps_l)/2 sum_{j!=k} N_j, lM = log M (so a fully underflowed density gives -inf exactly as
 * there).  lp_B_g_F (C,H,3) and p_Bt_g_Ft (C,U,3) are written only when non-NULL (the fit path
 * never reads them; the Python mirror exposes them lazily). */
int fcd_lik_tables(fcd_ctx *ctx, const double *b, const double *bt, int64_t C, int64_t H, int64_t U,
                   const double *theta12_host, double *S_B, double *lM, double *lp_B_g_F,
                   double *p_Bt_g_Ft, fcd_stream stream);

/* ---- forward sampler: UnsharedRegionModel.sample, fcdiff/model.py:52-236, on the device ---------------
 * Counter RNG (Philox), all variables drawn in parallel; the reference's MT19937 stream is not reproduced (the host
 * sampler of the Python mirror does that) -- same distribution.  Type INDICES are returned: r (Nreg,U), t (C,U),
 * f (C,), f_tilde (C,U) uint8; b (C,H), b_tilde (C,U) float64 clipped to [-1,1].  Edges in the fitter's order. */
int fcd_model_sample(fcd_ctx *ctx, const double *theta12_host, int64_t Nreg, int64_t H, int64_t U, uint64_t seed, uint8_t *r,
                     uint8_t *t, uint8_t *f, uint8_t *f_tilde, double *b, double *b_tilde, fcd_stream stream);

/* ---- front-end: region x time series -> edge-major correlations --------------------------------------
 * Not in the reference (its inputs are already correlations, fit.py:20-23); oracle = numpy.corrcoef.
 * ts (S, Nreg, T) -> out (C, S), out[c][s] = corrcoef(ts[s])[n, m] for c = n(n-1)/2 + m, n > m: the layout of
 * b / bt.  fp64 MFMA Gram product per subject.  fisher_z != 0 applies atanh; keep it 0 with the reference's model
 * defaults, which are calibrated on raw correlations clipped to [-1, 1] (fcdiff/model.py:213, 236). */
int fcd_corr_edges(fcd_ctx *ctx, const double *ts, int64_t S, int64_t Nreg, int64_t T, int fisher_z, double *out,
                   fcd_stream stream);

/* ---- variational updates ------------------------------------------------------------------ */
/* UnsharedRegionFit._update_lq_F, fit.py:157-174 (+ _eval_q_R_w 382-406). */
int fcd_vb_update_qF(fcd_ctx *ctx, const double *lq_R, const double *S_B, const double *lM,
                     const double *hyper, int64_t Nreg, int64_t U, double *lq_F, fcd_stream stream);
/* UnsharedRegionFit._update_lq_R, fit.py:176-198: Gauss-Seidel over regions, in place. */
int fcd_vb_update_qR(fcd_ctx *ctx, const double *lq_F, const double *lM, const double *hyper,
                     int64_t Nreg, int64_t U, int edge_mode, double *lq_R, fcd_stream stream);
/* The six terms of _eval_energy, fit.py:142-155 / 447-539, in its order:
 * terms6 = {E[ln p(f)], E[ln p(b|f)], E[ln p(r)], E[ln p(bt|f,r)], E[ln q_F], E[ln q_R]} (device).
 * energy = -t0 -t1 -t2 -t3 +t4 +t5.  Deterministic (fixed reduction order). */
int fcd_vb_energy(fcd_ctx *ctx, const double *lq_F, const double *lq_R, const double *S_B,
                  const double *lM, const double *hyper, int64_t Nreg, int64_t U, double *terms6,
                  fcd_stream stream);
/* _update_pi + _update_gamma, fit.py:208-220: out4 = {mean q_R[:,:,1], mean_c q_F[c,0,:]} (device);
 * when hyper != NULL also stores their logs there (the theta step without leaving the device). */
int fcd_vb_theta_step(fcd_ctx *ctx, const double *lq_F, const double *lq_R, int64_t Nreg, int64_t U,
                      double *out4, double *hyper, fcd_stream stream);

/* ---- (eta, epsilon) step: objective and analytic gradient --------------------------------------------
 * The reference sketches a bounded minimisation of -E_lM over (eta, epsilon) (fit.py:222-286) that cannot run there
 * (fit.py:239 calls an undefined name); its derivative helpers are complete (fit.py:600-697).  For weights
 * W (C, U, 3, 3) >= 0 these return out3 = {S, dS/d eta, dS/d epsilon} (device), S = sum W[c,u,k,l] ln M_kl(bt_cu):
 *   W = q_F[c,k] w_l(c,u)            -> S = E_lM (fit.py:489-511), energy derivative = -dS  (variational fit)
 *   W = pooled chain counts          -> Monte-Carlo EM objective of the sampler
 * theta12_host as in fcd_lik_tables.  Deterministic. */
int fcd_theta_sub_weights_vb(fcd_ctx *ctx, const double *lq_F, const double *lq_R, int64_t Nreg, int64_t U, double *W,
                             fcd_stream stream);
/* W[c,u,k,l] (+)= number of this rank's chains with f_c = k and mixture case l at (c,u)  (accumulate != 0: add). */
int fcd_gibbs_pair_counts(fcd_ctx *ctx, const uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg, int64_t U,
                          int64_t G, int accumulate, double *W, fcd_stream stream);
int fcd_theta_sub_objective(fcd_ctx *ctx, const double *bt, const double *W, int64_t C, int64_t U,
                            const double *theta12_host, double *out3, fcd_stream stream);
/* The FULL theta_sub objective the reference intends but comments out (fit.py:232-237 bounds, :250-251 / :266-267 pack
 * and unpack of mu and sigma^2, :282 the E[ln p(b | f)] term):
 *   S = sum_{c,k} wF[c,k] sum_h ln N(b[c,h]; mu_k, sigma_k) + sum W[c,u,k,l] ln M_kl(bt[c,u]),   wF[c,k] = sum_l W[c,0,k,l]
 * out9 = {S, dS/d eta, dS/d epsilon, dS/d mu_0..2, dS/d (sigma^2)_0..2} (device): the true gradient of S in the
 * reference's own parametrisation (it packs sigma ** 2).  Forms: fit.py:542-569, 572-597 (without quirk Q8, which
 * lives in a helper the objective never calls), 709-733; doc/methods.rst:715-944.  b == NULL leaves the first sum out.
 * Deterministic. */
int fcd_theta_full_objective(fcd_ctx *ctx, const double *b, const double *bt, const double *W, int64_t C, int64_t H,
                             int64_t U, const double *theta12_host, double *out9, fcd_stream stream);

/* ---- many-chain collapsed Gibbs sampler ---------------------------------------------------
 * Build-defined (the reference ships only the variational fitter, doc/methods.rst:236-239).  Its two
 * conditionals are the reference's updates at one-hot q:
 *   p(f_c = k | r)      = softmax_k of fit.py:170-173 with q_R one-hot
 *   p(r_nu = j | f, r)  = softmax_j of fit.py:187-194 with q_F, q_R one-hot
 * Randomness: Philox4x32-10, key = seed, counter = (site, global chain id, sweep, kind), so a chain's
 * trajectory depends only on (seed, chain id): not on G, the launch geometry or the number of GPUs. */
int fcd_gibbs_state_size(int64_t Nreg, int64_t U, int64_t G, size_t *f_bytes, size_t *r_bytes);
/* f ~ Uniform{0,1,2}, r ~ Bernoulli(pi) from the counter RNG (kinds 0, 1). */
int fcd_gibbs_init(fcd_ctx *ctx, uint8_t *f_state, uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G,
                   int64_t chain0, uint64_t seed, double pi, fcd_stream stream);
/* Edge-major DIFFERENCE table for the f step: lMf (C, U, 3, 2), lMf[c][u][l][k-1] = lM[c,u,k,l] - lM[c,u,0,l]
 * (the log-odds of type k against type 0 contributed by patient u in mixture case l).  48*C*U bytes. */
int fcd_gibbs_edge_tables(fcd_ctx *ctx, const double *lM, int64_t Nreg, int64_t U, double *lMf, fcd_stream stream);
/* Redraw every f_c of every chain given r (edges are conditionally independent).  With lMf only the two
 * log-odds against type 0 are accumulated (one 16-byte read + two adds per term); lMf == NULL runs the
 * kernel that accumulates the three sums of fit.py:170-173 from lM directly. */
int fcd_gibbs_f_step(fcd_ctx *ctx, const double *S_B, const double *lM, const double *lMf, const double *hyper,
                     uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G,
                     int64_t chain0, uint64_t seed, int64_t sweep, fcd_stream stream);
/* Region-major DIFFERENCE table for the r step: lMd (U, Nreg, Nreg, 3, 2),
 *   lMd[u][n][m][k][t] = t ? lM[c,u,k,1] - lM[c,u,k,2] : lM[c,u,k,2] - lM[c,u,k,0],  c = edge(n, m),
 * the contribution of region m (state t) to s1 - s0 of fit.py:187-194 when f_c = k; edge() = the
 * ordered-pair edge id of `edge_mode` (zeros at m == n).  Built once per table build; 48*U*Nreg*Nreg bytes. */
int fcd_gibbs_region_tables(fcd_ctx *ctx, const double *lM, int64_t Nreg, int64_t U, int edge_mode, double *lMd,
                            fcd_stream stream);
/* Redraw every r_nu of every chain given f, regions in order 0..Nreg-1 (systematic scan; patients
 * and chains in parallel).  With lMd (made with the SAME edge_mode) the blocked path runs: panel kernels
 * stream lMd rows through LDS, small diagonal kernels resolve the in-order dependence (only s1 - s0
 * is formed, as a sum of table differences: same conditional, one add per term).  lMd == NULL
 * selects the generic kernel that gathers from lM directly (any shape, much slower). */
int fcd_gibbs_r_step(fcd_ctx *ctx, const double *lM, const double *lMd, const double *hyper,
                     const uint8_t *f_state, uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G,
                     int64_t chain0, uint64_t seed, int64_t sweep, int edge_mode, fcd_stream stream);
/* n_sweeps x (f step, r step), sweeps numbered sweep0, sweep0+1, ...  When counts != NULL the pooled
 * statistics of the LAST sweep are stored there (see fcd_gibbs_stats). */
int fcd_gibbs_sweeps(fcd_ctx *ctx, const double *S_B, const double *lM, const double *lMf, const double *lMd,
                     const double *hyper, uint8_t *f_state, uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G, int64_t chain0,
                     uint64_t seed, int64_t sweep0, int64_t n_sweeps, int edge_mode, int64_t *counts,
                     fcd_stream stream);
/* Pooled sufficient statistics over the G chains (the all-reduce payload):
 * counts[0..7] = {sum r, #f=0, #f=1, #f=2, G, 0, 0, 0} (int64, device, overwritten). */
int fcd_gibbs_stats(fcd_ctx *ctx, const uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg, int64_t U,
                    int64_t G, int64_t *counts, fcd_stream stream);
/* M-step for (pi, gamma) from pooled counts (after the cross-GPU all-reduce): the sample version of
 * fit.py:208-220.  Writes hyper[0..4]; pi is kept inside [1/(2n), 1-1/(2n)], n = G*Nreg*U sites. */
int fcd_gibbs_mstep(fcd_ctx *ctx, const int64_t *counts, int64_t Nreg, int64_t U, double *hyper,
                    fcd_stream stream);
/* Running marginal counts over sweeps AND chains: cnt_f (C,3) += [f_c = k], cnt_r (Nreg,U) += r_nu
 * (uint32, device); posterior marginals = counts / (sweeps * G). */
int fcd_gibbs_accumulate(fcd_ctx *ctx, const uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg,
                         int64_t U, int64_t G, uint32_t *cnt_f, uint32_t *cnt_r, fcd_stream stream);
/* fcd_gibbs_stats and fcd_gibbs_accumulate in ONE pass over the state and ONE launch (the pooled sums are kept in
 * accumulators of the context that the kernel itself puts back to zero: no memset):
 * counts (nullable) is overwritten as by fcd_gibbs_stats; cnt_f / cnt_r (both or neither) are incremented. */
int fcd_gibbs_tally(fcd_ctx *ctx, const uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg, int64_t U,
                    int64_t G, int64_t *counts, uint32_t *cnt_f, uint32_t *cnt_r, fcd_stream stream);
/* The sampler loop of one rank between two exchanges of pooled statistics -- what the fit loop calls:
 *   for i in 0 .. n_sweeps-1:   f pass, r pass (sweep number sweep0 + i), then ONE tally launch that
 *     - adds the state to the marginal counters cnt_f / cnt_r (nullable, both or neither) when sweep0 + i >= accumulate_from,
 *     - when mstep_every > 0 and (i + 1) % mstep_every == 0 runs the (pi, gamma) M-step of fcd_gibbs_mstep on THIS
 *       rank's pooled counts and writes hyper (single-rank use; with several ranks pass 0, all-reduce `counts` and
 *       call fcd_gibbs_mstep),
 *     - makes the packed r words of the next f pass.
 *   counts (nullable) receives the pooled statistics of the LAST sweep.
 * 4 launches per sweep (f pass; packing, which also carries the f half of the tally in workgroups of its own; pipelined r
 * pass; the rest of the tally: r counts, slot words, M-step) where the pipelined r pass fits the device at once, else
 * ceil(Nreg/16) + 4 (one launch per block step). */
int fcd_gibbs_run(fcd_ctx *ctx, const double *S_B, const double *lM, const double *lMf, const double *lMd,
                  double *hyper,
                  uint8_t *f_state, uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G, int64_t chain0, uint64_t seed,
                  int64_t sweep0, int64_t n_sweeps, int edge_mode, int64_t mstep_every, int64_t accumulate_from,
                  int64_t *counts, uint32_t *cnt_f, uint32_t *cnt_r, fcd_stream stream);
/* log p(f, r, b, bt; theta) of each chain = minus the first four terms of fit.py:149-152 at one-hot q.
 * out (G,) doubles. */
int fcd_gibbs_logjoint(fcd_ctx *ctx, const double *S_B, const double *lM, const double *hyper,
                       const uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G,
                       double *out, fcd_stream stream);
/* Number of anomalous (region, patient) sites of each chain, sum_{n,u} r_nu: out (G,) uint32 (overwritten).  The second
 * scalar of the chain diagnostics (split R-hat / ESS), next to the log-joint. */
int fcd_gibbs_chain_rsum(fcd_ctx *ctx, const uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G, uint32_t *out,
                         fcd_stream stream);
/* Unnormalised conditional log-weights of EVERY site given the current state, nothing updated:
 * cond_f (G, C, 3), cond_r (G, Nreg, U, 2).  Either may be NULL.  (Parity hook + Rao-Blackwell use.) */
int fcd_gibbs_conditionals(fcd_ctx *ctx, const double *S_B, const double *lM, const double *hyper,
                           const uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg, int64_t U,
                           int64_t G, int edge_mode, double *cond_f, double *cond_r, fcd_stream stream);
/* Plain views of the packed state: f (G, C) uint8, r (G, Nreg, U) uint8. */
int fcd_gibbs_export_state(fcd_ctx *ctx, const uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg,
                           int64_t U, int64_t G, uint8_t *f, uint8_t *r, fcd_stream stream);
int fcd_gibbs_import_state(fcd_ctx *ctx, const uint8_t *f, const uint8_t *r, int64_t Nreg, int64_t U,
                           int64_t G, uint8_t *f_state, uint64_t *r_bits, fcd_stream stream);
/* 53-bit uniforms of the counter RNG for a list of (idx, chain, sweep, kind) counters:
 * out[2*i + half].  Lets a host binding check its own Philox against the device's. */
int fcd_philox_uniforms(fcd_ctx *ctx, const uint32_t *ctr4, int64_t n, uint64_t seed, double *out,
                        fcd_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* FCDIFF_HIP_H */
