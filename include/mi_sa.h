/*
 * mi_sa.h -- C ABI of the MI355X simulated-annealing engine (libmi_sa.so).
 *
 * This is the drop-in boundary for the anneal ("sampler") step of
 * michal7kw/scRNA_seq_QAnnealing_Clustering.  The reference has NO native FFI for this path: its
 * sampler calls are Python method calls on third-party D-Wave objects
 *     sampler.sample_qubo(Q, ...)      /root/reference/Python_Functions/BQM_clustering.py:57,75,85,245,263,273
 *     sampler.sample(bqm, ...)         /root/reference/Python_Functions/BQM_clustering.py:386
 *     LeapHybridDQMSampler().sample_dqm(dqm, ...)   /root/reference/Python_Functions/DQM_clustering.py:45
 * The nearest native analogue is dwave-neal's C entry point `general_simulated_annealing(states,
 * energies, num_samples, h, coupler_starts, coupler_ends, coupler_weights, sweeps_per_beta,
 * beta_schedule, seed, ...)` (un-vendored third party).  Each entry point
 * below names the reference call it serves.  The Python class `MI355XSampler`
 * (scrna_seq_qannealing_clustering_amd/sampler.py) binds these with ctypes and presents the dimod
 * Sampler surface; INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; every array is caller-allocated HOST memory unless stated otherwise;
 *     the library stages inputs into HBM once per problem and keeps results in HBM until fetched.
 *   - return 0 (MI_OK) or a negative MI_E* code; mi_last_error() gives a thread-local message.
 *   - one internal HIP stream per problem handle; calls on one handle are serialised by the caller.
 *   - replica r of a run has GLOBAL id (replica_offset + r); the random stream of a replica depends
 *     only on (seed, global id), so sharding replicas over GPUs does not change any replica's result.
 *   - binary states are uint8 0/1, R x n row-major; Potts labels are uint16, R x n row-major.
 */
#ifndef MI_SA_H
#define MI_SA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK            0
#define MI_EINVAL       -1   /* bad argument */
#define MI_ENODEV       -2   /* no usable gfx950 device */
#define MI_EHIP         -3   /* HIP runtime error (see mi_last_error) */
#define MI_ENOMEM       -4
#define MI_EUNSUPPORTED -5   /* size / variant not built */
#define MI_ESTATE       -6   /* call out of order (e.g. fetch before anneal) */

#define MI_KIND_DENSE     1
#define MI_KIND_CSR_RANK1 2
#define MI_KIND_POTTS_CSR 3

typedef struct mi_sa_problem mi_sa_problem;

const char *mi_last_error(void);
int mi_abi_version(void);
int mi_device_count(int *out_count);
/* name (<= len-1 chars), CU count and HBM bytes of a device */
int mi_device_info(int device, char *name, int len, int *out_cus, uint64_t *out_hbm_bytes);

/* ---- problems ------------------------------------------------------------------------------ */

/* Dense binary model  E(x) = x^T Qs x + offset ; Qs n x n row-major SYMMETRIC fp32, diagonal =
 * linear terms.  Serves sampler.sample_qubo(Q) for an arbitrary QUBO dict (BQM_clustering.py:57;
 * QA_subsampling.py:42,56,65; other_tools.py:62) and sampler.sample(bqm) (BQM_clustering.py:386). */
int mi_sa_problem_create_dense_f32(const float *Qs, int n, double offset, int device,
                                   mi_sa_problem **out);

/* Structured binary model E(x) = sum lin_i x_i + sum_{i<j} (c_pair + S_ij) x_i x_j + offset,
 * S symmetric sparse in CSR (both directions stored).  This is exactly the shape of the reference's
 * graph-partition QUBO (BQM_clustering.py:38-47: sparse cut term + 2*gamma on every pair).  Rows of up to 4096
 * neighbours (up to 64 they are register resident); n <= 2^20. */
int mi_sa_problem_create_csr_rank1_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                                       const float *lin, float c_pair, int n, double offset,
                                       int device, mi_sa_problem **out);

/* Potts / DQM model E(l) = lin_offset + sum_{u<v, l_u==l_v} (c_pair + S_uv), K cases per variable:
 * the model clustering_dqm builds (DQM_clustering.py:29-43) and hands to sample_dqm (:45).  Rows of any
 * width up to 4096 neighbours (up to 64 they are register resident); n <= 40000, K <= 64. */
int mi_sa_problem_create_potts_csr_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                                       float c_pair, int n, int K, double lin_offset, int device,
                                       mi_sa_problem **out);

/* Optional fp64 coefficients for the REPORTED energies of a structured problem (csr_rank1 / potts_csr).
 * The chain always runs on the fp32 model given at creation; after this call the final energies
 * (mi_sa_fetch, mi_sa_best) are evaluated on the device from these values instead, i.e. in the caller's own
 * fp64 model -- what dimod's SampleSet.from_samples_bqm does on the host after the reference's sampler
 * calls (BQM_clustering.py:57,75,85; DQM_clustering.py:45).  val: one double per stored CSR entry, in the
 * order given at creation; lin: n doubles (csr_rank1; ignored for Potts, may be NULL). */
int mi_sa_problem_set_energy_model_f64(mi_sa_problem *p, const double *val, const double *lin, double c_pair);

int mi_sa_problem_destroy(mi_sa_problem *p);
int mi_sa_problem_info(const mi_sa_problem *p, int *kind, int *n, int *num_cases, int *device);

/* Tuning switches that never change results: "pace" (default 1) holds the replicas of one XCD together
 * at sweep boundaries so that their Q-row reads share that XCD's L2.  Kernel choice (all kernels of a model
 * run the same chain): "variant" (dense, n <= 4096: 0 auto, 1 wave per replica, 2 workgroup + LDS ring, 3 MFMA,
 * 4 scheduled), "mfma_permille" (default 600: chunks of a long dense run that accept at least this share go to
 * the MFMA kernel; 0 = never), "chunk_sweeps" (32), "k2_pair" (structured binary: 0 auto, 1 two replicas per
 * wavefront, 2 one), "k2_split" (the few-replica kernels: 0 auto = runs of up to "k2_split_max" (1024) replicas, 1 always when
 * eligible, 2 never; on models laid out in edge-free blocks of 128 / 256 seats "k2_wide" picks between ONE wavefront
 * sweeping a whole block per step (0 / 1, the default) and a workgroup of 2 / 4 wavefronts doing it (2)), "k2_tw" (0 / 1: the
 * random words and thresholds of a sweep come from a second "threshold" wavefront of the workgroup -- the pair kernel at 16
 * entries per variable and the one-wavefront few-replica kernel; 2: the sweeping wavefront computes them itself),
 * "k3_fast" (Potts: 0 the lean kernel csrc/potts_fast_kernels.hip when every slot is free of internal edges, K <= 16 and no
 * minimum size is set; 2 never), "xl_batched" (dense, n > 4096: 0 auto = all replicas together on the matrix cores from 256
 * replicas or n = 16384 up, 1 always, 2 a workgroup per replica), "xl_chain" (0 auto = the decisions and small passes of a group of eight blocks as one launch up to 512
 * replicas, 1 = one launch per block, 2 = fused always), "xl_chunk" (8) / "xl_cold_permille" (20): that batched kernel
 * hands a cooling run over to the per-replica kernel when a chunk of sweeps accepted less than this share; the decision
 * is taken per chunk on the host, by a worker thread of the problem ("xl_async" 1, the default: mi_sa_anneal returns at
 * once and the next call on the problem joins the worker, reporting its error if it had one; 0: in the calling thread).
 * One MODEL switch: "min_cluster_size" (Potts problems, default 0) -- every cluster keeps at least that many
 * members: a move out of a cluster holding exactly that many is rejected whatever its energy change.  This
 * is the `sum_i v[i][j] >= 20` constraint of the reference's CQM (CQM_clustering.py:46-48) as a hard
 * constraint; initial states must satisfy it. */
int mi_sa_set_option(mi_sa_problem *p, const char *key, long value);

/* Diagnostic: copies the first `words` pacing words of the last launch (layout in mi_sa.hip). */
int mi_sa_debug_pace(mi_sa_problem *p, unsigned int *out, int words);

/* Host-only planning (touches no device): the sweep order the structured kernels run fastest in.  Renumbers the
 * variables of a symmetric CSR graph so that the `slot` (= 64, the wavefront width) variables a wavefront sweeps
 * together are, as far as a degree-descending greedy balanced colouring manages, mutually NON-adjacent; inside
 * such a block the decisions of a sweep interact only through sum(x).  out_perm[new] = old.  Any visiting order
 * is a valid Metropolis sweep; the caller renumbers the model with it before mi_sa_problem_create_* and maps the
 * states back (scrna_seq_qannealing_clustering_amd/engine.py does).  Identity for n > 262144. */
int mi_sa_plan_slot_order(const int32_t *rowptr, const int32_t *col, int n, int slot, int64_t *out_perm);

/* The same planning with HOLES allowed: variable i gets the seat out_pos[i] = slot_index * slot + rank in a layout of
 * *out_slots slots (>= ceil(n / slot), at most max_slots), the fewest -- tried from the packed count upwards -- for which
 * the greedy pass leaves NO edge inside a slot; unused seats are holes.  Small or strongly clustered graphs (the
 * subgraphs the reference's recursive bisection solves, its 256-node benchmark graphs) cannot be packed into
 * ceil(n / 64) mutually non-adjacent blocks, and every block with an internal edge falls back to the serial accept
 * loop: ten partly filled blocks run 3x faster than five full ones.  If max_slots does not suffice the packed layout
 * is returned (*out_clashes = variables that share a slot with a neighbour, 0 otherwise; may be NULL).
 * The caller builds the padded model: a hole is a variable without couplings whose linear term is +infinity --
 * the structured kernels start such a variable at 0 and can never flip it (as the lanes past n). */
int mi_sa_plan_slot_layout(const int32_t *rowptr, const int32_t *col, int n, int slot, int max_slots,
                           int64_t *out_pos, int *out_slots, int *out_clashes);

/* Potts (k-way) models: marks the holes of a padded layout -- absent[i] != 0: no variable sits at position i (it must
 * have no couplings).  Such a position keeps label 0, belongs to no cluster (it is not counted in the cluster sizes
 * of the penalty term), takes no proposal.  Call before the first anneal of the problem. */
int mi_sa_problem_set_absent(mi_sa_problem *p, const uint8_t *absent);

/* Structured binary models: positive integer WEIGHTS of the uniform pair term,
 *     E(z) = offset + sum_i lin_i z_i + sum_{i<j} S_ij z_i z_j + c_pair sum_{i<j} w_i w_j z_i z_j
 * -- the shape a squared linear constraint with slack variables gives a QUBO, lam (sum_i x_i + sum_j c_j t_j - ub)^2:
 * what `clustering_bqm_3` builds through bqm.add_linear_inequality_constraint (BQM_clustering.py:373-380), w = 1 on the
 * cells and c_j on the slack bits, so that model runs on the structured kernels instead of the dense ones.
 * weights[i] >= 1 for every variable of the problem (the value at a hole is ignored).  The variables whose weight
 * is not 1 must lie inside ONE 64-variable slot that holds no variable of weight 1 (holes apart) and must have no
 * sparse couplings: MI_EINVAL otherwise.  Call before the first anneal of the problem. */
int mi_sa_problem_set_pair_weights(mi_sa_problem *p, const int32_t *weights);

/* Diagnostic: copies the first `words` (<= 16) 64-bit statistics words of the last run ([0..2] as in
 * mi_sa_fetch; [8..12] per-phase cycle sums of builds compiled with -DMI_K2_PROFILE, otherwise 0; with words = 16,
 * [14] / [15] = chunks of the last scheduled dense run served by the workgroup kernel / the MFMA kernel). */
int mi_sa_debug_stats(mi_sa_problem *p, uint64_t *out, int words);

/* ---- the anneal (replaces the sampler call itself) ------------------------------------------- */

/* R independent Metropolis chains x num_sweeps sweeps, one beta per sweep (betas[num_sweeps]).
 * init: NULL (random initial states from the replica's own stream) or R x n host states
 * (uint8 for binary kinds, uint16 labels for Potts).  resync_interval > 0 recomputes the cached
 * fp32 local fields from the state every that many sweeps (0 = never).  Asynchronous: returns after
 * enqueueing on the problem's stream; mi_sa_sync / mi_sa_fetch wait. */
int mi_sa_anneal(mi_sa_problem *p, int R, uint32_t replica_offset, int num_sweeps,
                 const double *betas, uint64_t seed, const void *init, int resync_interval);

/* Extended form used by parallel tempering and by resumable runs:
 *   sweep_offset          added to the sweep index in the random-number counter, so a run continued in
 *                         pieces draws exactly the numbers of one long run;
 *   MI_F_CONTINUE         start from the states the previous run left in HBM (same R; init must be NULL);
 *   MI_F_BETA_PER_REPLICA betas has R entries: replica r anneals at the constant beta betas[r] for all
 *                         num_sweeps sweeps of this call (one rung of a tempering ladder per replica). */
#define MI_F_CONTINUE         1u
#define MI_F_BETA_PER_REPLICA 2u
#define MI_F_TEMPS_RESIDENT   4u   /* betas ignored (may be NULL): every replica anneals at the temperature the
                                    * tempering state on the device holds for it (mi_sa_tempering_begin / _exchange) */
int mi_sa_anneal_ex(mi_sa_problem *p, int R, uint32_t replica_offset, int num_sweeps,
                    const double *betas, uint64_t seed, const void *init, int resync_interval,
                    uint32_t sweep_offset, uint32_t flags);

/* ---- parallel tempering: the replica-exchange step on the device (K6) --------------------------
 * A run has T x chains replicas; global replica g belongs to chain g / T and starts on ladder rung g % T.  This GPU
 * owns the R_local replicas first_replica .. (contiguous shard).  A round = mi_sa_anneal_ex(R_local, first_replica,
 * sweeps, NULL, seed, NULL, 0, round * sweeps, MI_F_TEMPS_RESIDENT | MI_F_CONTINUE (after the first round)), then
 * mi_sa_tempering_exchange(round, seed, all_energies): neighbouring rungs of each chain exchange with the Metropolis
 * rule on a counter-based random stream of (seed, round), rung INDICES move, states stay where they are.
 *   all_energies = NULL    one GPU owns every replica: the energies never leave HBM, nothing is copied either way;
 *   all_energies = host array of all T x chains energies in global order (the all-gather of the ranks' energies):
 *                          every rank runs the same exchange and gets the same rungs.
 * mi_sa_tempering_state copies the rung of every replica and the proposed / accepted exchange counts. */
int mi_sa_tempering_begin(mi_sa_problem *p, const double *ladder_betas, int T, int chains,
                          uint32_t first_replica, int R_local);
int mi_sa_tempering_exchange(mi_sa_problem *p, uint32_t round, uint64_t seed, const double *all_energies);
/* The same exchange with the all-gathered energies ALREADY IN HBM (d_all_energies: a device pointer on this problem's
 * GPU, T x chains doubles in global order -- the output buffer of the RCCL all-gather): nothing is staged through the
 * host.  The call returns after the exchange kernel has read the buffer. */
int mi_sa_tempering_exchange_dev(mi_sa_problem *p, uint32_t round, uint64_t seed, const double *d_all_energies);
int mi_sa_tempering_state(mi_sa_problem *p, int32_t *out_rung, uint64_t *out_proposed, uint64_t *out_accepted);

int mi_sa_sync(mi_sa_problem *p);

/* DEVICE pointers of the last run's result buffers (states R x n, energies R doubles) after waiting for it; they stay
 * valid until the next anneal on this handle with MORE replicas, or its destruction.  What a collective library needs
 * to send results GPU to GPU (the per-round all-gather of parallel tempering, distributed.gather_energies) without a
 * host copy.  Any of the out pointers may be NULL. */
int mi_sa_device_results(mi_sa_problem *p, void **out_d_states, double **out_d_energy, int *out_R);

/* Device time of the anneal kernel(s) of the last mi_sa_anneal on this handle, from HIP events
 * recorded on the problem's stream around the launch (milliseconds); implies mi_sa_sync. */
int mi_sa_last_kernel_ms(mi_sa_problem *p, float *out_ms);

/* Number of anneal-kernel launches that served the last mi_sa_anneal (long schedules are cut into launches
 * of `chunk_sweeps` sweeps whose state persists in HBM; results do not depend on the cut).  The time of
 * mi_sa_last_kernel_ms divided by this count is the average launch duration a profiler reports. */
int mi_sa_last_launch_count(mi_sa_problem *p, int *out_launches);

/* Name(s) of the kernel(s) those launches ran, as a profiler lists them (e.g. "k_anneal_csr_rank1_pair<16>";
 * several joined by " + " when a chunked dense run alternates kernels).  The library picks the kernel from the
 * model (kind, size, adjacency width, whether every slot is free of internal edges) and the number of replicas. */
int mi_sa_last_kernel_name(mi_sa_problem *p, char *out, int len);

/* Copy results of the last run to host: states (R x n, uint8 or uint16 by kind; nullable),
 * energies (R doubles, recomputed from the final state on device; nullable), stats (nullable):
 * stats[0] proposals, stats[1] accepted moves, stats[2] Q/CSR bytes read by accepted moves. */
int mi_sa_fetch(mi_sa_problem *p, void *out_states, double *out_energy, uint64_t *out_stats);

/* Best replica of the last run (reduced on device): the replica with the lowest fp64 energy (ties: the lowest
 * index) -- its local index, energy, and an order-preserving packed key
 * (sortable(float(E)) << 32) | global_replica_id  suitable for an integer MIN all-reduce across GPUs (RCCL has
 * no MINLOC; between GPUs the comparison therefore has fp32 resolution of E).  out_state (nullable) receives
 * that replica's n states. */
int mi_sa_best(mi_sa_problem *p, int *out_index, double *out_energy, uint64_t *out_key,
               void *out_state);

/* ---- several GPUs from one process (mi_multi_gpu_*) ------------------------------------
 * problems[d] = the SAME model created on device d (mi_sa_problem_create_*(…, device = d, …)).  The R_total replicas
 * with global ids replica_offset .. are sharded contiguously (remainders to the low devices), every device anneals
 * its shard on its own stream -- concurrently -- and because a replica's random stream is keyed by its GLOBAL id the
 * result does not depend on ndev.  _best: the lowest fp64 energy over the devices' own best replicas, ties to the lowest
 * global id -- the reduction the one-process-per-GPU path does with one RCCL MIN all-reduce
 * (distributed.global_best_f64); out_state receives the winner's n states from its owner.  _fetch: states / energies of all replicas in global order, stats
 * summed. */
int mi_multi_gpu_anneal(mi_sa_problem *const *problems, int ndev, int R_total, uint32_t replica_offset,
                        int num_sweeps, const double *betas, uint64_t seed, int resync_interval);
int mi_multi_gpu_best(mi_sa_problem *const *problems, int ndev, int *out_owner, uint32_t *out_global_id,
                      double *out_energy, void *out_state);
int mi_multi_gpu_fetch(mi_sa_problem *const *problems, int ndev, void *out_states, double *out_energy,
                       uint64_t *out_stats);

/* ---- one-shot conveniences (host in, host out) ---------------------------------------------- */

int mi_sa_qubo_dense_f32(const float *Qs, int n, double offset, int R, int num_sweeps,
                         const double *betas, uint64_t seed, const uint8_t *init,
                         uint8_t *out_states, double *out_energy, uint64_t *out_stats, int device);

/* Batched energy evaluation E_r = x_r^T Qs x_r + offset for R states (K4).  Serves SampleSet energy
 * re-evaluation (BQM_clustering.py:93-98, :133-146 read these energies).  path: 0 = auto (f32-input MFMA
 * when the batch is >= 32 states wide, i.e. a true dense contraction; exact-fp64 VALU otherwise),
 * 1 = VALU (every fp32 entry added once into fp64), 2 = MFMA (Qs SYMMETRIC, as everywhere in this header: only the
 * 128 x 128 blocks on and above the diagonal are multiplied, an off-diagonal block counts twice; fp32 partial
 * sums of <= 128 terms folded into fp64: ~1e-7 of sum|terms|).  out_kernel_ms (nullable): device time of the
 * evaluation kernels. */
int mi_energy_dense_f32(const float *Qs, int n, const uint8_t *X, int R, double offset,
                        double *out_energy, int device);
/* The same evaluation over an fp64 matrix (the caller's own coefficients, every entry added once into fp64):
 * the energies dimod's SampleSet.from_samples_bqm computes on the host for the samples of a dense model. */
int mi_energy_dense_f64(const double *Qs, int n, const uint8_t *X, int R, double offset,
                        double *out_energy, int device);
int mi_energy_dense_f32_ex(const float *Qs, int n, const uint8_t *X, int R, double offset,
                           double *out_energy, int device, int path, float *out_kernel_ms);

#ifdef __cplusplus
}
#endif
#endif /* MI_SA_H */
