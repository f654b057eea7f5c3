/*
 * oracle/sa_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object.  The product (scrna_seq_qannealing_clustering_amd/) never links, imports or
 * calls it; the product path fails loudly when its HIP library is missing.
 *
 * What is restated here, and from where:
 *
 *  (1) orc_sa_ising_neal  -- the simulated-annealing algorithm of dwave-neal
 *      (`neal.SimulatedAnnealingSampler`, dwave-neal 0.5.x / dwave-samplers >= 1.0), the local
 *      sampler a user of the reference substitutes for the D-Wave call at the boundary
 *      /root/reference/Python_Functions/BQM_clustering.py:57,75,85,245,263,273.  neal is a
 *      third-party dependency (requirements.txt:1 `dwave-ocean-sdk>=3.3.0`, a floor, not a pin)
 *      that is NOT vendored in /root/reference and NOT installable offline, so its published
 *      algorithm is restated from SURVEY.md section 8c ("sampler side"): Ising spins +-1,
 *      cached flip energies, variables visited in index order, the 44.36142/beta skip
 *      threshold, xorshift128+ consumed only on uphill proposals, energies recomputed at the
 *      end.  PARITY UNPINNED against real neal: the reference holds no output of any sampler
 *      (no tests, no captured SampleSet).  What pins this file is (a) the
 *      known-answer energies / exact optima of SURVEY.md section 8c, reproduced in
 *      tests/test_oracle_kat.py, and (b) brute force on small graphs.
 *
 *  (2) orc_sa_dense_philox / orc_sa_csr_rank1_philox / orc_potts_csr_philox -- the SAME Metropolis
 *      chain, restated with the counter-based Philox4x32-10 stream and fp32 cached local fields
 *      that the MI355X kernels use (DESIGN.md "Chain specification"), so that GPU and CPU agree
 *      flip for flip (bit-exact states).  Independent restatement: nothing in csrc/ is included.
 *
 *  (3) orc_energy_*  -- fp64 energy evaluators (E = x^T Qs x + offset) and the cut-edge counter.
 *  (4) orc_bruteforce_qubo -- exact optimum for n <= 26.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; fmaf only where written explicitly).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11).  Known-answer vectors checked in tests.               */
/* ------------------------------------------------------------------------------------------ */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Random word for (variable i, sweep s, global replica g, stream tag): DESIGN.md "RNG addressing".
 * Four consecutive 64-variable slots share one Philox block so that a 64-lane wave needs one
 * Philox evaluation per four slots. */
static inline uint32_t chain_word(uint64_t seed, uint32_t i, uint32_t s, uint32_t g, uint32_t tag)
{
    uint32_t ctr[4], key[2], out[4];
    ctr[0] = ((i >> 8) << 6) | (i & 63u);
    ctr[1] = s;
    ctr[2] = g;
    ctr[3] = tag;
    key[0] = (uint32_t)seed;
    key[1] = (uint32_t)(seed >> 32);
    orc_philox4x32_10(ctr, key, out);
    return out[(i >> 6) & 3u];
}

uint32_t orc_chain_word(uint64_t seed, uint32_t i, uint32_t s, uint32_t g, uint32_t tag)
{
    return chain_word(seed, i, s, g, tag);
}

/* -ln(u), u = 2 - m in (0,1], m in [1,2) built from the top 23 bits of r.  Degree-7 polynomial for
 * ln(1+t)/t on [sqrt(1/2)-1, sqrt(2)-1]; every operation is a single IEEE fp32 op or fmaf, so a
 * GPU evaluating the same sequence returns the same bits. */
float orc_neglog_u(uint32_t r)
{
    union { uint32_t u; float f; } cv;
    cv.u = 0x3f800000u | (r >> 9);
    float u = 2.0f - cv.f;                 /* exact */
    cv.f = u;
    int e = (int)(cv.u >> 23) - 127;
    cv.u = (cv.u & 0x007fffffu) | 0x3f800000u;
    float m = cv.f;
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float t = m - 1.0f;                    /* exact */
    float p = -0x1.9f9af6p-4f;
    p = fmaf(p, t, 0x1.4cd8dcp-3f);
    p = fmaf(p, t, -0x1.61491cp-3f);
    p = fmaf(p, t, 0x1.977bcp-3f);
    p = fmaf(p, t, -0x1.ff611p-3f);
    p = fmaf(p, t, 0x1.555a22p-2f);
    p = fmaf(p, t, -0x1.00007cp-1f);
    p = fmaf(p, t, 0x1.fffffep-1f);
    float lnm = p * t;
    return fmaf((float)(-e), 0x1.62e43p-1f, -lnm);   /* (float)(-e), not -(float)e: +0 at u = 1 (e = 0, lnm = 0) */
}

/* The same value with the range reduction written the way the device kernels do it (csrc/mi_sa_device.h
 * neglog_u): adding 2^23 - 0x3504f4 to the bits of u carries into the exponent field exactly when the mantissa
 * exceeds that of 1.41421356f.  orc_neglog_forms_differ() counts the inputs (of all 2^23) on which the two
 * forms return different bits: 0. */
float orc_neglog_u_carry(uint32_t r)
{
    union { uint32_t u; float f; } cv;
    cv.u = 0x3f800000u | (r >> 9);
    cv.f = 2.0f - cv.f;
    const uint32_t ub = cv.u;
    const int neg_e = 127 - (int)((ub + 0x004afb0cu) >> 23);
    cv.u = ub + ((uint32_t)neg_e << 23);
    const float t = cv.f - 1.0f;
    float p = -0x1.9f9af6p-4f;
    p = fmaf(p, t, 0x1.4cd8dcp-3f);
    p = fmaf(p, t, -0x1.61491cp-3f);
    p = fmaf(p, t, 0x1.977bcp-3f);
    p = fmaf(p, t, -0x1.ff611p-3f);
    p = fmaf(p, t, 0x1.555a22p-2f);
    p = fmaf(p, t, -0x1.00007cp-1f);
    p = fmaf(p, t, 0x1.fffffep-1f);
    return fmaf((float)neg_e, 0x1.62e43p-1f, -(p * t));
}

/* ... and with the reduction the pair kernel's packed form uses (csrc/mi_sa_device.h neglog_u2): t2 = the exponent
 * field of (bits(u) + C), in place; m = u times the power of two whose bits are 0x7f000000 - t2; the exponent as
 * fmaf((float)(int)t2, -2^-23, 127). */
float orc_neglog_u_scaled(uint32_t r)
{
    union { uint32_t u; float f; } cv, sc;
    cv.u = 0x3f800000u | (r >> 9);
    cv.f = 2.0f - cv.f;
    const uint32_t t2 = (cv.u + 0x004afb0cu) & 0x7f800000u;
    sc.u = 0x7f000000u - t2;
    const float m = cv.f * sc.f;
    const float neg_e = fmaf((float)(int)t2, -0x1p-23f, 127.0f);
    const float t = m - 1.0f;
    float p = -0x1.9f9af6p-4f;
    p = fmaf(p, t, 0x1.4cd8dcp-3f);
    p = fmaf(p, t, -0x1.61491cp-3f);
    p = fmaf(p, t, 0x1.977bcp-3f);
    p = fmaf(p, t, -0x1.ff611p-3f);
    p = fmaf(p, t, 0x1.555a22p-2f);
    p = fmaf(p, t, -0x1.00007cp-1f);
    p = fmaf(p, t, 0x1.fffffep-1f);
    return fmaf(neg_e, 0x1.62e43p-1f, -(p * t));
}

long orc_neglog_forms_differ(void)
{
    long bad = 0;
    for (uint32_t k = 0; k < (1u << 23); ++k) {
        const float a = orc_neglog_u(k << 9), b = orc_neglog_u_carry(k << 9), c = orc_neglog_u_scaled(k << 9);
        if (memcmp(&a, &b, sizeof a) != 0 || memcmp(&a, &c, sizeof a) != 0) ++bad;
    }
    return bad;
}

/* ------------------------------------------------------------------------------------------ */
/* (2a) dense fp32 chain, Philox stream -- mirror of kernel K1                                  */
/* ------------------------------------------------------------------------------------------ */
/* Qs: n x n row-major symmetric fp32, diagonal = linear terms.  E(x) = x^T Qs x + offset.
 * Local field f_i = Qs_ii + sum_{j != i} 2 Qs_ij x_j ; flip delta dE_i = (1 - 2 x_i) f_i.
 * temps[s] = (float)(1.0 / (double)betas[s]).  Proposal i at sweep s is accepted iff
 *     dE_i < neglog_u(word(i, s, g, 0)) * temps[s]        (fp32 multiply, fp32 compare)
 * Field (re)initialisation: f = diag; then for j ascending with x_j = 1: f_i += 2 Qs_ji (i != j).
 * Returns 0.  out_states R x n (uint8 0/1), out_energy R doubles (fp64 re-evaluation + offset),
 * out_stats[0] += proposals, out_stats[1] += accepted flips. */
static void field_init_dense(const float *Qs, int n, const uint8_t *x, float *f)
{
    for (int i = 0; i < n; ++i) f[i] = Qs[(size_t)i * n + i];
    for (int j = 0; j < n; ++j) {
        if (!x[j]) continue;
        const float *row = Qs + (size_t)j * n;
        for (int i = 0; i < n; ++i) {
            if (i == j) continue;
            float q2 = row[i] + row[i];
            f[i] = f[i] + q2;
        }
    }
}

int orc_sa_dense_philox(const float *Qs, int n, double offset, int R, uint32_t replica_offset,
                        int num_sweeps, const double *betas, uint64_t seed, const uint8_t *init,
                        int resync_interval, uint8_t *out_states, double *out_energy,
                        uint64_t *out_stats, uint32_t sweep_offset, int betas_per_replica)
{
    /* sweep_offset: added to the sweep index in the RNG counter (a run continued in pieces draws the
     * numbers of one long run).  betas_per_replica: betas has R entries, replica r anneals at betas[r]. */
    uint64_t tot_prop = 0, tot_acc = 0;
    const int nb = betas_per_replica ? R : num_sweeps;
    float *temps = (float *)malloc(sizeof(float) * (size_t)(nb > 0 ? nb : 1));
    for (int s = 0; s < nb; ++s) temps[s] = (float)(1.0 / betas[s]);
#pragma omp parallel for schedule(dynamic) reduction(+ : tot_prop, tot_acc)
    for (int r = 0; r < R; ++r) {
        uint32_t g = replica_offset + (uint32_t)r;
        uint8_t *x = out_states + (size_t)r * n;
        float *f = (float *)malloc(sizeof(float) * (size_t)n);
        if (init) memcpy(x, init + (size_t)r * n, (size_t)n);
        else
            for (int i = 0; i < n; ++i) x[i] = (uint8_t)(chain_word(seed, (uint32_t)i, 0, g, 1) >> 31);
        field_init_dense(Qs, n, x, f);
        for (int s = 0; s < num_sweeps; ++s) {
            float T = temps[betas_per_replica ? r : s];
            for (int i = 0; i < n; ++i) {
                float thr = orc_neglog_u(chain_word(seed, (uint32_t)i, (uint32_t)s + sweep_offset, g, 0)) * T;
                float dE = x[i] ? -f[i] : f[i];
                ++tot_prop;
                if (dE < thr) {
                    float sgn = x[i] ? -1.0f : 1.0f;
                    const float *row = Qs + (size_t)i * n;
                    for (int j = 0; j < n; ++j) {
                        if (j == i) continue;
                        float q2 = row[j] + row[j];
                        f[j] = f[j] + sgn * q2;
                    }
                    x[i] ^= 1;
                    ++tot_acc;
                }
            }
            if (resync_interval > 0 && (s + 1) % resync_interval == 0) field_init_dense(Qs, n, x, f);
        }
        /* fp64 energy from scratch */
        double E = 0.0;
        for (int i = 0; i < n; ++i) {
            if (!x[i]) continue;
            const float *row = Qs + (size_t)i * n;
            double acc = 0.0;
            for (int j = 0; j < n; ++j)
                if (x[j]) acc += (double)row[j];
            E += acc;
        }
        out_energy[r] = E + offset;
        free(f);
    }
    free(temps);
    if (out_stats) { out_stats[0] += tot_prop; out_stats[1] += tot_acc; }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* (2b) CSR + uniform pair term chain -- mirror of kernel K2                                    */
/* ------------------------------------------------------------------------------------------ */
/* Model: E(x) = sum_i lin_i x_i + sum_{i<j} (c + S_ij) x_i x_j + offset where S is sparse and
 * symmetric (CSR: rowptr/col/val hold S_ij for every stored neighbour, both directions) and c is
 * the uniform pair coefficient (2*gamma for BQM_clustering.py:46-47).  Local field
 *     f_i = lin_i + sum_j S_ij x_j + c (s - x_i),   s = sum x.
 * Chain: variables are visited in index order, in blocks of 64 consecutive variables ("slots", the
 * wavefront width of the kernel).  When a slot is entered, the sparse part of the field of each of its
 * variables is evaluated FRESH from the current state, fp32, in stored CSR order:
 *     g_i = lin_i ;  for e in row i:  if x[col_e]:  g_i = g_i + val_e
 * Inside the slot, an accepted flip of i updates the g of its neighbours IN THE SAME SLOT (g_j += sgn*S_ij,
 * one fp32 add each, in flip order) and the integer s;  f_i = g_i + c * (float)(s - x_i)  (one fp32
 * multiply, one add).  No field survives a slot, so nothing drifts and there is nothing to re-synchronise
 * (resync_interval is accepted and ignored).
 * A position whose linear term is +infinity is a HOLE (a seat of a padded sweep layout that holds no variable,
 * include/mi_sa.h: mi_sa_plan_slot_layout): it is 0 from the start -- an explicit initial state included -- and
 * takes no proposal; the random numbers stay addressed by position, so the holes only shift them. */
#define ORC_SLOT 64
/* wgt (nullable): positive integer WEIGHTS a_i of the pair term,
 *     E(z) = offset + sum_i lin_i z_i + sum_{i<j} S_ij z_i z_j + c sum_{i<j} a_i a_j z_i z_j
 * -- the shape of a squared linear constraint with slack variables, lam (sum_i x_i + sum_j c_j t_j - ub)^2
 * (BQM_clustering.py:373-380 through dimod's add_linear_inequality_constraint: a_i = 1 on the cells, c_j on the
 * slack bits).  The chain carries A = sum_j a_j z_j as an integer:  f_i = g_i + (c * (float)a_i) * (float)(A - a_i z_i)
 * (fp32: one multiply for the coefficient, one multiply and one add, no contraction); with a_i = 1 everywhere
 * (wgt == NULL) this is chain (2b) as it always was, bit for bit. */
int orc_sa_csr_rank1_philox_w(const int *rowptr, const int *col, const float *val, const float *lin,
                              float c_pair, int n, double offset, int R, uint32_t replica_offset,
                              int num_sweeps, const double *betas, uint64_t seed, const uint8_t *init,
                              int resync_interval, uint8_t *out_states, double *out_energy,
                              uint64_t *out_stats, uint32_t sweep_offset, int betas_per_replica, const int *wgt)
{
    uint64_t tot_prop = 0, tot_acc = 0;
    const int nb = betas_per_replica ? R : num_sweeps;
    float *temps = (float *)malloc(sizeof(float) * (size_t)(nb > 0 ? nb : 1));
    for (int s = 0; s < nb; ++s) temps[s] = (float)(1.0 / betas[s]);
#pragma omp parallel for schedule(dynamic) reduction(+ : tot_prop, tot_acc)
    for (int r = 0; r < R; ++r) {
        uint32_t gid = replica_offset + (uint32_t)r;
        uint8_t *x = out_states + (size_t)r * n;
        float g[ORC_SLOT];
        long S = 0;                                                   /* sum_j a_j x_j */
        if (init) memcpy(x, init + (size_t)r * n, (size_t)n);
        else
            for (int i = 0; i < n; ++i) x[i] = (uint8_t)(chain_word(seed, (uint32_t)i, 0, gid, 1) >> 31);
        for (int i = 0; i < n; ++i) if (isinf(lin[i])) x[i] = 0;      /* a hole of a padded layout (see above) */
        for (int i = 0; i < n; ++i) S += x[i] ? (wgt ? wgt[i] : 1) : 0;
        for (int s = 0; s < num_sweeps; ++s) {
            float T = temps[betas_per_replica ? r : s];
            for (int i0 = 0; i0 < n; i0 += ORC_SLOT) {
                const int i1 = i0 + ORC_SLOT < n ? i0 + ORC_SLOT : n;
                for (int i = i0; i < i1; ++i) {
                    float gi = lin[i];
                    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e)
                        if (x[col[e]]) gi = gi + val[e];
                    g[i - i0] = gi;
                }
                for (int i = i0; i < i1; ++i) {
                    if (isinf(lin[i])) continue;                      /* hole: no variable sits here, nothing is proposed */
                    const int ai = wgt ? wgt[i] : 1;
                    float thr = orc_neglog_u(chain_word(seed, (uint32_t)i, (uint32_t)s + sweep_offset, gid, 0)) * T;
                    float cw = c_pair * (float)ai;                    /* (a_i = 1: c_pair itself) */
                    float fi = g[i - i0] + cw * (float)(S - (x[i] ? ai : 0));
                    float dE = x[i] ? -fi : fi;
                    ++tot_prop;
                    if (dE < thr) {
                        float sgn = x[i] ? -1.0f : 1.0f;
                        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e)
                            if (col[e] >= i0 && col[e] < i1) g[col[e] - i0] = g[col[e] - i0] + sgn * val[e];
                        S += x[i] ? -ai : ai;
                        x[i] ^= 1;
                        ++tot_acc;
                    }
                }
            }
        }
        (void)resync_interval;
        double E = 0.0;
        double A = 0.0, A2 = 0.0;                                     /* sum a_i x_i, sum a_i^2 x_i (exact in fp64) */
        for (int i = 0; i < n; ++i) {
            if (!x[i]) continue;
            const double ai = wgt ? (double)wgt[i] : 1.0;
            A += ai;
            A2 += ai * ai;
            E += (double)lin[i];
            double acc = 0.0;
            for (int e = rowptr[i]; e < rowptr[i + 1]; ++e)
                if (x[col[e]]) acc += (double)val[e];
            E += 0.5 * acc;
        }
        E += (double)c_pair * 0.5 * (A * A - A2);                     /* (a = 1: cnt (cnt - 1) / 2) */
        out_energy[r] = E + offset;
    }
    free(temps);
    if (out_stats) { out_stats[0] += tot_prop; out_stats[1] += tot_acc; }
    return 0;
}

int orc_sa_csr_rank1_philox(const int *rowptr, const int *col, const float *val, const float *lin,
                            float c_pair, int n, double offset, int R, uint32_t replica_offset,
                            int num_sweeps, const double *betas, uint64_t seed, const uint8_t *init,
                            int resync_interval, uint8_t *out_states, double *out_energy,
                            uint64_t *out_stats, uint32_t sweep_offset, int betas_per_replica)
{
    return orc_sa_csr_rank1_philox_w(rowptr, col, val, lin, c_pair, n, offset, R, replica_offset, num_sweeps, betas, seed,
                                     init, resync_interval, out_states, out_energy, out_stats, sweep_offset,
                                     betas_per_replica, NULL);
}

/* How often does fp32 chain arithmetic decide differently from fp64 (neal computes in doubles)?  Runs chain (2b) exactly
 * as above (the fp32 decisions drive the trajectory) and, for every proposal, also evaluates the same predicate in
 * fp64 from the same state and the same random word:  g64 = lin + sum S_ij x_j,  f64 = g64 + c (s - x_i),
 * accept64 iff +-f64 < -ln(u) / beta  with the natural logarithm of libm.  counts[0] += proposals,
 * counts[1] += proposals on which the two predicates disagree, counts[2] += accepted (fp32). */
int orc_csr_rank1_fp32_vs_fp64_decisions(const int *rowptr, const int *col, const float *val, const float *lin,
                                         float c_pair, int n, int R, uint32_t replica_offset, int num_sweeps,
                                         const double *betas, uint64_t seed, uint64_t *counts)
{
    uint64_t prop = 0, differ = 0, acc = 0;
#pragma omp parallel for schedule(dynamic) reduction(+ : prop, differ, acc)
    for (int r = 0; r < R; ++r) {
        const uint32_t gid = replica_offset + (uint32_t)r;
        uint8_t *x = (uint8_t *)malloc((size_t)n);
        float g[ORC_SLOT];
        double g64[ORC_SLOT];
        int S = 0;
        for (int i = 0; i < n; ++i) { x[i] = (uint8_t)(chain_word(seed, (uint32_t)i, 0, gid, 1) >> 31); S += x[i]; }
        for (int s = 0; s < num_sweeps; ++s) {
            const float T = (float)(1.0 / betas[s]);
            for (int i0 = 0; i0 < n; i0 += ORC_SLOT) {
                const int i1 = i0 + ORC_SLOT < n ? i0 + ORC_SLOT : n;
                for (int i = i0; i < i1; ++i) {
                    float gi = lin[i];
                    double gd = (double)lin[i];
                    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e)
                        if (x[col[e]]) { gi = gi + val[e]; gd += (double)val[e]; }
                    g[i - i0] = gi;
                    g64[i - i0] = gd;
                }
                for (int i = i0; i < i1; ++i) {
                    const uint32_t w = chain_word(seed, (uint32_t)i, (uint32_t)s, gid, 0);
                    const float thr = orc_neglog_u(w) * T;
                    const float fi = g[i - i0] + c_pair * (float)(S - (int)x[i]);
                    const float dE = x[i] ? -fi : fi;
                    const int a32 = dE < thr;
                    union { uint32_t u; float f; } cv;
                    cv.u = 0x3f800000u | (w >> 9);
                    const double u = 2.0 - (double)cv.f;
                    const double f64 = g64[i - i0] + (double)c_pair * (double)(S - (int)x[i]);
                    const int a64 = (x[i] ? -f64 : f64) < -log(u) / betas[s];
                    ++prop;
                    differ += (uint64_t)(a32 != a64);
                    if (a32) {
                        const float sgn = x[i] ? -1.0f : 1.0f;
                        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e)
                            if (col[e] >= i0 && col[e] < i1) {
                                g[col[e] - i0] = g[col[e] - i0] + sgn * val[e];
                                g64[col[e] - i0] += (double)sgn * (double)val[e];
                            }
                        S += x[i] ? -1 : 1;
                        x[i] ^= 1;
                        ++acc;
                    }
                }
            }
        }
        free(x);
    }
    counts[0] += prop; counts[1] += differ; counts[2] += acc;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* (2c) Potts / DQM chain on CSR + uniform pair term -- mirror of kernel K3                     */
/* ------------------------------------------------------------------------------------------ */
/* Model (DQM_clustering.py:29-43 after its set_ overwrites):
 *     E(l) = lin_offset + sum_{u<v, l_u == l_v} B_uv,  B_uv = c + S_uv  (S sparse: -2w - c on edges)
 * State l_i in [0,K).  Proposal for variable i at sweep s: target label
 *     b = (a + 1 + (word(i,s,g,2) mod (K-1))) mod K      (uniform over the K-1 other labels)
 * dE = [h_i(b) + c*cnt_b] - [h_i(a) + c*(cnt_a - 1)],  h_i(q) = sum_{j in N(i), l_j == q} S_ij
 * accepted iff dE < neglog_u(word(i,s,g,0)) * temps[s].  h_i(b) - h_i(a) is recomputed per proposal from the
 * CSR row as one signed sum in stored order (fp32 adds / subtracts in that order); cnt are integers. */
/* min_size > 0 restricts the chain to labelings in which every cluster keeps at least min_size members
 * (the "cluster_size >= 20" constraints of CQM_clustering.py:46-48 as a hard constraint): a move out of a
 * cluster that holds exactly min_size variables is rejected whatever its dE.  The restricted chain still
 * satisfies detailed balance on the feasible set; the initial state must be feasible. */
/* absent (nullable): absent[i] != 0 -- position i is a HOLE of a padded sweep layout (include/mi_sa.h:
 * mi_sa_problem_set_absent): label 0, in no cluster, no proposal; random numbers stay addressed by position. */
int orc_potts_csr_philox_absent(const int *rowptr, const int *col, const float *val, float c_pair, int n,
                                int K, double lin_offset, int R, uint32_t replica_offset, int num_sweeps,
                                const double *betas, uint64_t seed, const uint16_t *init,
                                uint16_t *out_labels, double *out_energy, uint64_t *out_stats,
                                uint32_t sweep_offset, int betas_per_replica, int min_size, const uint8_t *absent)
{
    uint64_t tot_prop = 0, tot_acc = 0;
    const int nb = betas_per_replica ? R : num_sweeps;
    float *temps = (float *)malloc(sizeof(float) * (size_t)(nb > 0 ? nb : 1));
    for (int s = 0; s < nb; ++s) temps[s] = (float)(1.0 / betas[s]);
#pragma omp parallel for schedule(dynamic) reduction(+ : tot_prop, tot_acc)
    for (int r = 0; r < R; ++r) {
        uint32_t gid = replica_offset + (uint32_t)r;
        uint16_t *l = out_labels + (size_t)r * n;
        int *cnt = (int *)calloc((size_t)K, sizeof(int));
        if (init) memcpy(l, init + (size_t)r * n, sizeof(uint16_t) * (size_t)n);
        else
            for (int i = 0; i < n; ++i)
                l[i] = (uint16_t)(chain_word(seed, (uint32_t)i, 0, gid, 1) % (uint32_t)K);
        for (int i = 0; i < n; ++i) {
            if (absent && absent[i]) l[i] = 0;
            else cnt[l[i]]++;
        }
        for (int s = 0; s < num_sweeps; ++s) {
            float T = temps[betas_per_replica ? r : s];
            for (int i = 0; i < n && K > 1; ++i) {
                if (absent && absent[i]) continue;
                int a = l[i];
                int b = (a + 1 + (int)(chain_word(seed, (uint32_t)i, (uint32_t)s + sweep_offset, gid, 2) % (uint32_t)(K - 1))) % K;
                /* field difference h_i(b) - h_i(a) as ONE signed fp32 sum in stored order: + S_ij for a neighbour
                 * with the target label, - S_ij for one with the variable's own label (round 3; the difference of
                 * two separate sums before: the same number up to rounding, one table lookup and one fma per
                 * neighbour on the device, csrc/potts_fast_kernels.hip) */
                float hd = 0.0f;
                for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
                    int lj = l[col[e]];
                    if (lj == b) hd = hd + val[e];
                    else if (lj == a) hd = hd - val[e];
                }
                /* dE = (h_b + c cnt_b) - (h_a + c (cnt_a - 1)), evaluated as ONE fused multiply-add of the
                 * (exact, integer) size difference onto the (fp32) field difference -- chain specification
                 * 2c in DESIGN.md section 3; the device kernel evaluates the same expression */
                float dE = fmaf(c_pair, (float)(cnt[b] - (cnt[a] - 1)), hd);
                float thr = orc_neglog_u(chain_word(seed, (uint32_t)i, (uint32_t)s + sweep_offset, gid, 0)) * T;
                ++tot_prop;
                if (dE < thr && cnt[a] - 1 >= min_size) {
                    l[i] = (uint16_t)b;
                    cnt[a]--;
                    cnt[b]++;
                    ++tot_acc;
                }
            }
        }
        double E = 0.0;
        for (int i = 0; i < n; ++i)
            for (int e = rowptr[i]; e < rowptr[i + 1]; ++e)
                if (col[e] > i && l[col[e]] == l[i]) E += (double)val[e];
        for (int q = 0; q < K; ++q) E += (double)c_pair * 0.5 * (double)cnt[q] * (double)(cnt[q] - 1);
        out_energy[r] = E + lin_offset;
        free(cnt);
    }
    free(temps);
    if (out_stats) { out_stats[0] += tot_prop; out_stats[1] += tot_acc; }
    return 0;
}

int orc_potts_csr_philox_min(const int *rowptr, const int *col, const float *val, float c_pair, int n,
                             int K, double lin_offset, int R, uint32_t replica_offset, int num_sweeps,
                             const double *betas, uint64_t seed, const uint16_t *init,
                             uint16_t *out_labels, double *out_energy, uint64_t *out_stats,
                             uint32_t sweep_offset, int betas_per_replica, int min_size)
{
    return orc_potts_csr_philox_absent(rowptr, col, val, c_pair, n, K, lin_offset, R, replica_offset, num_sweeps, betas,
                                       seed, init, out_labels, out_energy, out_stats, sweep_offset, betas_per_replica,
                                       min_size, NULL);
}

int orc_potts_csr_philox(const int *rowptr, const int *col, const float *val, float c_pair, int n,
                         int K, double lin_offset, int R, uint32_t replica_offset, int num_sweeps,
                         const double *betas, uint64_t seed, const uint16_t *init,
                         uint16_t *out_labels, double *out_energy, uint64_t *out_stats,
                         uint32_t sweep_offset, int betas_per_replica)
{
    return orc_potts_csr_philox_min(rowptr, col, val, c_pair, n, K, lin_offset, R, replica_offset, num_sweeps, betas,
                                    seed, init, out_labels, out_energy, out_stats, sweep_offset, betas_per_replica, 0);
}

/* ------------------------------------------------------------------------------------------ */
/* (1) neal restatement: Ising, fp64, xorshift128+, CSR adjacency                               */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint64_t s0, s1; } xs128p_t;

static inline uint64_t xs128p_next(xs128p_t *st)
{
    uint64_t x = st->s0;
    uint64_t const y = st->s1;
    st->s0 = y;
    x ^= x << 23;
    st->s1 = x ^ y ^ (x >> 17) ^ (y >> 26);
    return st->s1 + y;
}

/* states: num_samples x n int8 (+-1), in/out.  Adjacency in CSR (nbr_ptr, nbr, nbr_J): for every
 * coupler (u,v,J) both u->v and v->u are stored, as neal builds its per-variable neighbour lists.
 * beta_schedule has num_betas entries, each held for sweeps_per_beta sweeps.  One RNG stream is
 * shared by all samples, which run sequentially (neal is single-threaded; reads are sequential).
 * Returns the number of samples completed.  stats: [0] proposals, [1] accepted flips. */
int orc_sa_ising_neal(int8_t *states, double *energies, int num_samples, int n, const double *h,
                      const int *nbr_ptr, const int *nbr, const double *nbr_J, int sweeps_per_beta,
                      const double *beta_schedule, int num_betas, uint64_t seed,
                      uint64_t *out_stats)
{
    const double RANDMAX = (double)0xFFFFFFFFFFFFFFFFull;
    xs128p_t rng;
    rng.s0 = seed ? seed : 0xFFFFFFFFFFFFFFFFull;
    rng.s1 = 0;
    uint64_t prop = 0, acc = 0;
    double *dE = (double *)malloc(sizeof(double) * (size_t)n);
    for (int smp = 0; smp < num_samples; ++smp) {
        int8_t *st = states + (size_t)smp * n;
        for (int v = 0; v < n; ++v) {
            double en = h[v];
            for (int e = nbr_ptr[v]; e < nbr_ptr[v + 1]; ++e) en += (double)st[nbr[e]] * nbr_J[e];
            dE[v] = -2.0 * (double)st[v] * en;
        }
        for (int bi = 0; bi < num_betas; ++bi) {
            double beta = beta_schedule[bi];
            for (int sw = 0; sw < sweeps_per_beta; ++sw) {
                double threshold = 44.36142 / beta;
                for (int v = 0; v < n; ++v) {
                    ++prop;
                    if (dE[v] >= threshold) continue;
                    int flip = 0;
                    if (dE[v] <= 0.0) flip = 1;
                    else {
                        uint64_t rnd = xs128p_next(&rng);
                        if (exp(-dE[v] * beta) * RANDMAX > (double)rnd) flip = 1;
                    }
                    if (flip) {
                        double mult = 4.0 * (double)st[v];
                        for (int e = nbr_ptr[v]; e < nbr_ptr[v + 1]; ++e)
                            dE[nbr[e]] += mult * nbr_J[e] * (double)st[nbr[e]];
                        st[v] = (int8_t)(-st[v]);
                        dE[v] = -dE[v];
                        ++acc;
                    }
                }
            }
        }
        double E = 0.0;
        for (int v = 0; v < n; ++v) {
            E += h[v] * (double)st[v];
            for (int e = nbr_ptr[v]; e < nbr_ptr[v + 1]; ++e)
                if (nbr[e] > v) E += (double)st[v] * nbr_J[e] * (double)st[nbr[e]];
        }
        energies[smp] = E;
    }
    free(dE);
    if (out_stats) { out_stats[0] += prop; out_stats[1] += acc; }
    return num_samples;
}

/* Same algorithm specialised to a DENSE coupling matrix (every variable neighbours every other,
 * as for BQM_clustering.py:46-47): J is n x n row-major symmetric with zero diagonal.  Used as the
 * timed CPU baseline on the dense PBMC QUBO (no neighbour-index indirection, so it is at least as
 * fast as neal's own list-of-vectors loop).  `threads` > 1 runs samples in parallel, each sample
 * with its own xorshift stream seeded seed + sample (a documented deviation from neal's single
 * stream, used only for the all-cores baseline). */
int orc_sa_ising_neal_dense(int8_t *states, double *energies, int num_samples, int n,
                            const double *h, const double *J, int sweeps_per_beta,
                            const double *beta_schedule, int num_betas, uint64_t seed, int threads,
                            uint64_t *out_stats)
{
    const double RANDMAX = (double)0xFFFFFFFFFFFFFFFFull;
    uint64_t prop = 0, acc = 0;
    xs128p_t shared;
    shared.s0 = seed ? seed : 0xFFFFFFFFFFFFFFFFull;
    shared.s1 = 0;
    if (threads < 1) threads = 1;
#ifdef _OPENMP
    omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic) reduction(+ : prop, acc) if (threads > 1)
    for (int smp = 0; smp < num_samples; ++smp) {
        xs128p_t local;
        xs128p_t *rng = &shared;
        if (threads > 1) {
            local.s0 = (seed + (uint64_t)smp) ? (seed + (uint64_t)smp) : 0xFFFFFFFFFFFFFFFFull;
            local.s1 = 0;
            rng = &local;
        }
        int8_t *st = states + (size_t)smp * n;
        double *dE = (double *)malloc(sizeof(double) * (size_t)n);
        for (int v = 0; v < n; ++v) {
            double en = h[v];
            const double *row = J + (size_t)v * n;
            for (int u = 0; u < n; ++u) en += (double)st[u] * row[u];
            dE[v] = -2.0 * (double)st[v] * en;
        }
        for (int bi = 0; bi < num_betas; ++bi) {
            double beta = beta_schedule[bi];
            for (int sw = 0; sw < sweeps_per_beta; ++sw) {
                double threshold = 44.36142 / beta;
                for (int v = 0; v < n; ++v) {
                    ++prop;
                    if (dE[v] >= threshold) continue;
                    int flip = 0;
                    if (dE[v] <= 0.0) flip = 1;
                    else {
                        uint64_t rnd = xs128p_next(rng);
                        if (exp(-dE[v] * beta) * RANDMAX > (double)rnd) flip = 1;
                    }
                    if (flip) {
                        double mult = 4.0 * (double)st[v];
                        const double *row = J + (size_t)v * n;
                        double keep = dE[v];
                        for (int u = 0; u < n; ++u) dE[u] += mult * row[u] * (double)st[u];
                        st[v] = (int8_t)(-st[v]);
                        dE[v] = -keep;
                        ++acc;
                    }
                }
            }
        }
        double E = 0.0;
        for (int v = 0; v < n; ++v) {
            E += h[v] * (double)st[v];
            const double *row = J + (size_t)v * n;
            double a = 0.0;
            for (int u = v + 1; u < n; ++u) a += row[u] * (double)st[u];
            E += (double)st[v] * a;
        }
        energies[smp] = E;
        free(dE);
    }
    if (out_stats) { out_stats[0] += prop; out_stats[1] += acc; }
    return num_samples;
}

/* ------------------------------------------------------------------------------------------ */
/* (3) evaluators                                                                               */
/* ------------------------------------------------------------------------------------------ */
void orc_energy_dense_f64(const float *Qs, int n, const uint8_t *X, int R, double offset,
                          double *out)
{
    for (int r = 0; r < R; ++r) {
        const uint8_t *x = X + (size_t)r * n;
        double E = 0.0;
        for (int i = 0; i < n; ++i) {
            if (!x[i]) continue;
            const float *row = Qs + (size_t)i * n;
            double acc = 0.0;
            for (int j = 0; j < n; ++j)
                if (x[j]) acc += (double)row[j];
            E += acc;
        }
        out[r] = E + offset;
    }
}

void orc_energy_dense_f64d(const double *Qs, int n, const uint8_t *X, int R, double offset,
                           double *out)
{
    for (int r = 0; r < R; ++r) {
        const uint8_t *x = X + (size_t)r * n;
        double E = 0.0;
        for (int i = 0; i < n; ++i) {
            if (!x[i]) continue;
            const double *row = Qs + (size_t)i * n;
            double acc = 0.0;
            for (int j = 0; j < n; ++j)
                if (x[j]) acc += row[j];
            E += acc;
        }
        out[r] = E + offset;
    }
}

/* number of edges (eu[k], ev[k]) whose endpoints carry different labels: the integer edge cut */
void orc_cut_edges_u8(const int *eu, const int *ev, int m, const uint8_t *X, int n, int R,
                      int64_t *out)
{
    for (int r = 0; r < R; ++r) {
        const uint8_t *x = X + (size_t)r * n;
        int64_t c = 0;
        for (int k = 0; k < m; ++k) c += (x[eu[k]] != x[ev[k]]);
        out[r] = c;
    }
}

void orc_cut_edges_u16(const int *eu, const int *ev, int m, const uint16_t *L, int n, int R,
                       int64_t *out)
{
    for (int r = 0; r < R; ++r) {
        const uint16_t *l = L + (size_t)r * n;
        int64_t c = 0;
        for (int k = 0; k < m; ++k) c += (l[eu[k]] != l[ev[k]]);
        out[r] = c;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* (4) exact optimum by Gray-code enumeration, n <= 26; Qs symmetric fp64                       */
/* ------------------------------------------------------------------------------------------ */
int orc_bruteforce_qubo(const double *Qs, int n, double offset, double *min_E, uint64_t *argmin,
                        uint64_t *num_min, double *second_E)
{
    if (n < 1 || n > 26) return -1;
    double f[26];
    int x[26];
    for (int i = 0; i < n; ++i) { f[i] = Qs[(size_t)i * n + i]; x[i] = 0; }
    double E = 0.0, best = 0.0, second = INFINITY;
    uint64_t bestmask = 0, nbest = 1, mask = 0;
    const double tol = 1e-9;
    uint64_t total = 1ull << n;
    for (uint64_t k = 1; k < total; ++k) {
        int i = __builtin_ctzll(k);
        double d = x[i] ? -1.0 : 1.0;
        E += d * f[i];
        x[i] ^= 1;
        mask ^= (1ull << i);
        for (int j = 0; j < n; ++j)
            if (j != i) f[j] += d * 2.0 * Qs[(size_t)i * n + j];
        if (E < best - tol) {
            second = best; best = E; bestmask = mask; nbest = 1;
        } else if (fabs(E - best) <= tol) {
            ++nbest;
            if (mask < bestmask) bestmask = mask;
        } else if (E < second - tol) {
            second = E;
        }
    }
    *min_E = best + offset;
    *argmin = bestmask;
    *num_min = nbest;
    *second_E = second + offset;
    return 0;
}
