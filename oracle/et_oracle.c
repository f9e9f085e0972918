/*
 * CPU restatement of the reference's native EigenTrust path in plain C -- TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (protocol_amd/) never links or calls it.
 *
 * What it restates (paths relative to /root/reference):
 *   eigentrust-zk/src/circuits/dynamic_sets/native.rs:234-283  filter_peers_ops
 *   eigentrust-zk/src/circuits/dynamic_sets/native.rs:286-337  converge
 * in two shapes:
 *   oracle_dense_converge_fr   the reference's own dense loop nest (same loops, same order),
 *                              runtime-sized instead of const-generic;
 *   oracle_sparse_converge_*   the same mathematics over an edge list, for sizes a dense
 *                              N x N matrix cannot hold.  A peer whose filtered row sums to
 *                              zero contributes s[j]/(n_valid-1) to every other valid peer
 *                              (native.rs:263-278); that dense row of ones is evaluated as
 *                              (D - dangling_i*s[i]) * inv(n_valid-1), D = sum of dangling s[j].
 *
 * Field arithmetic: halo2curves 0.3.2 bn256::Fr (Cargo.lock:1768-1770) is not part of the
 * reference tree.  Results are observable only as canonical little-endian 32-byte integers
 * modulo r (eigentrust-zk/src/utils.rs:327-329), so any correct arithmetic modulo
 * r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
 * (eigentrust-zk/src/params/rns/bn256.rs:6) yields the same bytes.  Here: 4 x 64-bit limbs,
 * Montgomery CIOS with unsigned __int128.
 *
 * Pinned by tests/test_oracle.py against oracle/eigentrust_ref.py (Python integers), the
 * committed golden vectors and the reference's CLI asset.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;           /* Montgomery form, value < r */

static const uint64_t MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL,
                                0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t R2M[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL,
                                0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
static const uint64_t N0 = 0xc2e1f593efffffffULL;   /* -r^-1 mod 2^64 */

static int geq_mod(const uint64_t a[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > MOD[i]) return 1;
        if (a[i] < MOD[i]) return 0;
    }
    return 1;
}
static void sub_mod(uint64_t a[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a[i] - MOD[i] - br;
        a[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
static fe fe_add(fe a, fe b) {
    fe c; u128 cy = 0;
    for (int i = 0; i < 4; ++i) { u128 t = (u128)a.l[i] + b.l[i] + cy; c.l[i] = (uint64_t)t; cy = t >> 64; }
    if (cy || geq_mod(c.l)) sub_mod(c.l);
    return c;
}
static fe fe_sub(fe a, fe b) {
    fe c; u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a.l[i] - b.l[i] - br; c.l[i] = (uint64_t)d; br = (d >> 64) & 1; }
    if (br) { u128 cy = 0; for (int i = 0; i < 4; ++i) { u128 t = (u128)c.l[i] + MOD[i] + cy; c.l[i] = (uint64_t)t; cy = t >> 64; } }
    return c;
}
static fe fe_mul(fe a, fe b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 cy = 0;
        for (int j = 0; j < 4; ++j) { u128 x = (u128)a.l[j] * b.l[i] + t[j] + cy; t[j] = (uint64_t)x; cy = x >> 64; }
        u128 x = (u128)t[4] + cy; t[4] = (uint64_t)x; t[5] = (uint64_t)(x >> 64);
        uint64_t m = t[0] * N0;
        cy = ((u128)m * MOD[0] + t[0]) >> 64;
        for (int j = 1; j < 4; ++j) { u128 y = (u128)m * MOD[j] + t[j] + cy; t[j - 1] = (uint64_t)y; cy = y >> 64; }
        x = (u128)t[4] + cy; t[3] = (uint64_t)x; t[4] = t[5] + (uint64_t)(x >> 64);
    }
    fe c; memcpy(c.l, t, 32);
    if (t[4] || geq_mod(c.l)) sub_mod(c.l);
    return c;
}
static int fe_is_zero(fe a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
static fe fe_zero(void) { fe z = {{0, 0, 0, 0}}; return z; }
static fe fe_from_raw(const uint64_t raw[4]) { fe a, r2; memcpy(a.l, raw, 32); memcpy(r2.l, R2M, 32); return fe_mul(a, r2); }
static void fe_to_raw(fe a, uint64_t raw[4]) { fe one = {{1, 0, 0, 0}}; fe c = fe_mul(a, one); memcpy(raw, c.l, 32); }
static fe fe_from_u64(uint64_t v) { uint64_t raw[4] = {v, 0, 0, 0}; return fe_from_raw(raw); }
static fe fe_from_u128(uint64_t lo, uint64_t hi) { uint64_t raw[4] = {lo, hi, 0, 0}; return fe_from_raw(raw); }
/* a^(r-2); 0 -> 0, the reference's invert().unwrap_or(ZERO) (native.rs:308) */
static fe fe_inv(fe a) {
    uint64_t e[4]; memcpy(e, MOD, 32); e[0] -= 2;
    fe acc = fe_from_u64(1);
    for (int i = 255; i >= 0; --i) {
        acc = fe_mul(acc, acc);
        if ((e[i / 64] >> (i % 64)) & 1) acc = fe_mul(acc, a);
    }
    return acc;
}
static int le_bytes_to_fe(const uint8_t *b, fe *out) {
    uint64_t raw[4]; memcpy(raw, b, 32);
    if (geq_mod(raw)) return -1;
    *out = fe_from_raw(raw); return 0;
}
static void fe_to_le_bytes(fe a, uint8_t *b) { uint64_t raw[4]; fe_to_raw(a, raw); memcpy(b, raw, 32); }

/* ---- exported field known-answer hooks (32-byte canonical little-endian in and out) ---- */
int oracle_fr_mul(const uint8_t *a, const uint8_t *b, uint8_t *out) { fe x, y; if (le_bytes_to_fe(a, &x) || le_bytes_to_fe(b, &y)) return -1; fe_to_le_bytes(fe_mul(x, y), out); return 0; }
int oracle_fr_add(const uint8_t *a, const uint8_t *b, uint8_t *out) { fe x, y; if (le_bytes_to_fe(a, &x) || le_bytes_to_fe(b, &y)) return -1; fe_to_le_bytes(fe_add(x, y), out); return 0; }
int oracle_fr_sub(const uint8_t *a, const uint8_t *b, uint8_t *out) { fe x, y; if (le_bytes_to_fe(a, &x) || le_bytes_to_fe(b, &y)) return -1; fe_to_le_bytes(fe_sub(x, y), out); return 0; }
int oracle_fr_inv(const uint8_t *a, uint8_t *out) { fe x; if (le_bytes_to_fe(a, &x)) return -1; fe_to_le_bytes(fe_inv(x), out); return 0; }

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/*
 * Dense restatement, same loop nest as native.rs:286-337.
 *   member[i]   != 0 where slot i holds a peer (set[i].0 != 0)
 *   has_row[i]  != 0 where ops holds a row for slot i's address (update_op was called)
 *   ops         n*n canonical field elements, row i = what Opinion::validate returned for peer i
 * Returns 0, -1 insufficient peers (native.rs:289), -2 conservation failure (native.rs:334),
 * -3 non-canonical input.  *loop_seconds receives the time of the iteration loop (:319-329).
 */
int oracle_dense_converge_fr(uint32_t n, uint32_t num_iter, uint64_t init_lo, uint64_t init_hi,
                             const uint8_t *member, const uint8_t *has_row, const uint8_t *ops,
                             uint8_t *out, double *loop_seconds) {
    uint32_t valid = 0;
    for (uint32_t i = 0; i < n; ++i) valid += member[i] != 0;
    if (valid < 2) return -1;
    fe *m = (fe *)calloc((size_t)n * n, sizeof(fe));
    fe one = fe_from_u64(1);
    /* filter_peers_ops :234-283 and the row assembly :292-302 */
    for (uint32_t i = 0; i < n; ++i) {
        if (!member[i]) continue;                       /* empty slot -> zero row */
        fe sum = fe_zero();
        for (uint32_t j = 0; j < n; ++j) {
            fe v = fe_zero();
            if (has_row[i] && member[j] && j != i) {
                if (le_bytes_to_fe(ops + ((size_t)i * n + j) * 32, &v)) { free(m); return -3; }
            }
            m[(size_t)i * n + j] = v;
            sum = fe_add(sum, v);
        }
        if (fe_is_zero(sum))
            for (uint32_t j = 0; j < n; ++j) if (member[j] && j != i) m[(size_t)i * n + j] = one;
    }
    /* normalise :305-314 */
    for (uint32_t i = 0; i < n; ++i) {
        fe sum = fe_zero();
        for (uint32_t j = 0; j < n; ++j) sum = fe_add(sum, m[(size_t)i * n + j]);
        fe inv = fe_inv(sum);
        for (uint32_t j = 0; j < n; ++j) m[(size_t)i * n + j] = fe_mul(m[(size_t)i * n + j], inv);
    }
    fe init = fe_from_u128(init_lo, init_hi);
    fe *s = (fe *)calloc(n, sizeof(fe)), *ns = (fe *)calloc(n, sizeof(fe));
    fe sum_initial = fe_zero();
    for (uint32_t i = 0; i < n; ++i) { s[i] = member[i] ? init : fe_zero(); sum_initial = fe_add(sum_initial, s[i]); }
    double t0 = now_s();
    for (uint32_t it = 0; it < num_iter; ++it) {        /* :319-329 */
        for (uint32_t i = 0; i < n; ++i) {
            fe acc = fe_zero();
            for (uint32_t j = 0; j < n; ++j) acc = fe_add(fe_mul(m[(size_t)j * n + i], s[j]), acc);
            ns[i] = acc;
        }
        memcpy(s, ns, (size_t)n * sizeof(fe));
    }
    if (loop_seconds) *loop_seconds = now_s() - t0;
    fe sum_final = fe_zero();
    for (uint32_t i = 0; i < n; ++i) { sum_final = fe_add(sum_final, s[i]); fe_to_le_bytes(s[i], out + (size_t)i * 32); }
    int rc = memcmp(sum_initial.l, sum_final.l, 32) == 0 ? 0 : -2;
    free(m); free(s); free(ns);
    return rc;
}

/* ---------------------------------------------------------------------------------------
 * Sparse restatement.  Input is an attestation list (from, to, value); a later entry for the
 * same (from, to) pair replaces an earlier one, as attestation_matrix[origin][dest] = Some(att)
 * does in eigentrust/src/lib.rs:398-416.
 * ------------------------------------------------------------------------------------- */
typedef struct { uint32_t from, to; uint64_t seq; } ekey;
static int cmp_ft(const void *a, const void *b) {
    const ekey *x = (const ekey *)a, *y = (const ekey *)b;
    if (x->from != y->from) return x->from < y->from ? -1 : 1;
    if (x->to != y->to) return x->to < y->to ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq);
}

typedef struct {
    uint32_t n, n_valid;
    uint64_t nnz;             /* surviving edges of non-dangling peers */
    uint64_t *rowptr;         /* CSR of C^T: row = destination, n+1 entries */
    uint32_t *src;            /* source peer of each in-edge, ascending inside a row */
    fe *w_fr;                 /* normalised weight, Montgomery form */
    double *w_f64;
    long double *w_ld;
    uint8_t *dangling;        /* member whose filtered row sums to zero in the field */
    const uint8_t *member;
} sparse_graph;

static void sg_free(sparse_graph *g) { free(g->rowptr); free(g->src); free(g->w_fr); free(g->w_f64); free(g->w_ld); free(g->dangling); }

/* value_fr != NULL: 32-byte canonical values; else value_u32. */
static int sg_build(sparse_graph *g, uint32_t n, const uint8_t *member, uint64_t n_edges,
                    const uint32_t *from, const uint32_t *to, const uint32_t *value_u32,
                    const uint8_t *value_fr, int want_fr, int want_f64) {
    memset(g, 0, sizeof(*g));
    g->n = n; g->member = member;
    for (uint32_t i = 0; i < n; ++i) g->n_valid += member[i] != 0;
    if (g->n_valid < 2) return -1;                       /* native.rs:288-289 */
    /* filter: drop self-opinions and opinions from/about empty slots (native.rs:248-260) */
    ekey *k = (ekey *)malloc((n_edges ? n_edges : 1) * sizeof(ekey));
    uint64_t m = 0;
    for (uint64_t e = 0; e < n_edges; ++e) {
        uint32_t f = from[e], t = to[e];
        if (f >= n || t >= n) { free(k); return -4; }
        if (!member[f] || !member[t] || f == t) continue;
        k[m].from = f; k[m].to = t; k[m].seq = e; ++m;
    }
    {
        /* lists that arrive grouped by attester and ordered by rated peer (what a client that stores a matrix
         * writes out) need no sort: at 10^8 edges the sort is most of this function */
        int sorted = 1;
        for (uint64_t e = 1; e < m && sorted; ++e) sorted = cmp_ft(&k[e - 1], &k[e]) < 0;
        if (!sorted) qsort(k, m, sizeof(ekey), cmp_ft);
    }
    uint64_t u = 0;                                       /* last write wins */
    for (uint64_t e = 0; e < m; ++e) {
        if (e + 1 < m && k[e + 1].from == k[e].from && k[e + 1].to == k[e].to) continue;
        k[u++] = k[e];
    }
    m = u;
    /* per-source sums in the field (native.rs:263) and as integers (native.rs:368) */
    fe *sum_fr = (fe *)calloc(n, sizeof(fe));
    long double *sum_ld = (long double *)calloc(n, sizeof(long double));
    fe *val = (fe *)malloc((m ? m : 1) * sizeof(fe));
    long double *val_ld = (long double *)malloc((m ? m : 1) * sizeof(long double));
    for (uint64_t e = 0; e < m; ++e) {
        uint64_t q = k[e].seq;
        if (value_fr) {
            if (le_bytes_to_fe(value_fr + q * 32, &val[e])) { free(k); free(sum_fr); free(sum_ld); free(val); free(val_ld); return -3; }
            uint64_t raw[4]; memcpy(raw, value_fr + q * 32, 32);
            val_ld[e] = (long double)raw[0] + 18446744073709551616.0L * ((long double)raw[1] + 18446744073709551616.0L * ((long double)raw[2] + 18446744073709551616.0L * (long double)raw[3]));
        } else {
            val[e] = fe_from_u64(value_u32[q]);
            val_ld[e] = (long double)value_u32[q];
        }
        sum_fr[k[e].from] = fe_add(sum_fr[k[e].from], val[e]);
        sum_ld[k[e].from] += val_ld[e];
    }
    g->dangling = (uint8_t *)calloc(n, 1);
    fe *inv = (fe *)calloc(n, sizeof(fe));
    {
        /* one field inversion for all rows (prefix products), same values as inverting each sum */
        fe *pre = (fe *)malloc((size_t)n * sizeof(fe));
        fe run = fe_from_u64(1);
        for (uint32_t i = 0; i < n; ++i) {
            if (!member[i]) continue;
            if (fe_is_zero(sum_fr[i])) { g->dangling[i] = 1; continue; }   /* native.rs:264: row of ones */
            pre[i] = run;
            run = fe_mul(run, sum_fr[i]);
        }
        fe back = fe_inv(run);
        for (uint32_t i = n; i-- > 0;) {
            if (!member[i] || g->dangling[i]) continue;
            inv[i] = fe_mul(back, pre[i]);
            back = fe_mul(back, sum_fr[i]);
        }
        free(pre);
    }
    /* transpose to CSR by destination; edges of dangling sources are replaced by the row of ones */
    g->rowptr = (uint64_t *)calloc((size_t)n + 2, sizeof(uint64_t));
    for (uint64_t e = 0; e < m; ++e) if (!g->dangling[k[e].from]) g->rowptr[k[e].to + 1]++;
    for (uint32_t i = 0; i < n; ++i) g->rowptr[i + 1] += g->rowptr[i];
    g->nnz = g->rowptr[n];
    g->src = (uint32_t *)malloc((g->nnz ? g->nnz : 1) * sizeof(uint32_t));
    if (want_fr) g->w_fr = (fe *)malloc((g->nnz ? g->nnz : 1) * sizeof(fe));
    if (want_f64) { g->w_f64 = (double *)malloc((g->nnz ? g->nnz : 1) * sizeof(double)); g->w_ld = (long double *)malloc((g->nnz ? g->nnz : 1) * sizeof(long double)); }
    uint64_t *cur = (uint64_t *)malloc(((size_t)n + 1) * sizeof(uint64_t));
    memcpy(cur, g->rowptr, ((size_t)n + 1) * sizeof(uint64_t));
    for (uint64_t e = 0; e < m; ++e) {                    /* k is sorted by from: sources ascend per row */
        uint32_t f = k[e].from;
        if (g->dangling[f]) continue;
        uint64_t p = cur[k[e].to]++;
        g->src[p] = f;
        if (want_fr) g->w_fr[p] = fe_mul(val[e], inv[f]);  /* native.rs:312 */
        if (want_f64) { g->w_ld[p] = val_ld[e] / sum_ld[f]; g->w_f64[p] = (double)val_ld[e] / (double)sum_ld[f]; }
    }
    free(cur); free(k); free(sum_fr); free(sum_ld); free(val); free(val_ld); free(inv);
    return 0;
}

/*
 * Fr-exact sparse converge.  out: n canonical 32-byte scores (empty slots 0, native.rs:317).
 * n_threads <= 1 runs the scalar loop.  Returns 0 / -1 / -2 / -3 / -4 as above.
 */
int oracle_sparse_converge_fr(uint32_t n, const uint8_t *member, uint64_t n_edges,
                              const uint32_t *from, const uint32_t *to, const uint32_t *value_u32,
                              const uint8_t *value_fr, uint32_t num_iter, uint64_t init_lo,
                              uint64_t init_hi, int n_threads, uint8_t *out, uint64_t *nnz_out,
                              double *loop_seconds) {
    sparse_graph g;
    int rc = sg_build(&g, n, member, n_edges, from, to, value_u32, value_fr, 1, 0);
    if (rc) return rc;
    if (nnz_out) *nnz_out = g.nnz;
    fe init = fe_from_u128(init_lo, init_hi);
    fe inv_nm1 = fe_inv(fe_from_u64(g.n_valid - 1));
    fe *s = (fe *)calloc(n, sizeof(fe)), *ns = (fe *)calloc(n, sizeof(fe));
    fe sum_initial = fe_zero();
    for (uint32_t i = 0; i < n; ++i) { s[i] = member[i] ? init : fe_zero(); sum_initial = fe_add(sum_initial, s[i]); }
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    double t0 = now_s();
    for (uint32_t it = 0; it < num_iter; ++it) {
        fe D = fe_zero();
        for (uint32_t j = 0; j < n; ++j) if (g.dangling[j]) D = fe_add(D, s[j]);
#pragma omp parallel for schedule(dynamic, 1024) if (n_threads > 1)
        for (uint32_t i = 0; i < n; ++i) {
            fe acc = fe_zero();
            if (member[i]) {
                for (uint64_t p = g.rowptr[i]; p < g.rowptr[i + 1]; ++p) acc = fe_add(fe_mul(g.w_fr[p], s[g.src[p]]), acc);
                fe d = g.dangling[i] ? fe_sub(D, s[i]) : D;
                acc = fe_add(acc, fe_mul(d, inv_nm1));
            }
            ns[i] = acc;
        }
        fe *tmp = s; s = ns; ns = tmp;
    }
    if (loop_seconds) *loop_seconds = now_s() - t0;
    fe sum_final = fe_zero();
    for (uint32_t i = 0; i < n; ++i) { sum_final = fe_add(sum_final, s[i]); fe_to_le_bytes(s[i], out + (size_t)i * 32); }
    rc = memcmp(sum_initial.l, sum_final.l, 32) == 0 ? 0 : -2;
    free(s); free(ns); sg_free(&g);
    return rc;
}

/*
 * Floating-point sparse converge: out_f64 in double arithmetic, out_ld (optional) the same
 * loop in long double as the accuracy reference at sizes exact rationals cannot reach.
 * new = (1-alpha)*(C^T s + dangling term) + alpha*p[i]; p == NULL means s0/sum(s0)*sum(s0) = s0.
 */
int oracle_sparse_converge_f64(uint32_t n, const uint8_t *member, uint64_t n_edges,
                               const uint32_t *from, const uint32_t *to, const uint32_t *value_u32,
                               const uint8_t *value_fr, uint32_t num_iter, double initial,
                               double alpha, const double *pretrust, int n_threads,
                               double *out_f64, long double *out_ld, double *residual_l1,
                               uint64_t *nnz_out, double *loop_seconds) {
    sparse_graph g;
    int rc = sg_build(&g, n, member, n_edges, from, to, value_u32, value_fr, 0, 1);
    if (rc) return rc;
    if (nnz_out) *nnz_out = g.nnz;
    double *s = (double *)calloc(n, sizeof(double)), *ns = (double *)calloc(n, sizeof(double));
    for (uint32_t i = 0; i < n; ++i) s[i] = member[i] ? initial : 0.0;
    double inv_nm1 = 1.0 / (double)(g.n_valid - 1), res = 0.0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    double t0 = now_s();
    for (uint32_t it = 0; it < num_iter; ++it) {
        double D = 0.0;
        for (uint32_t j = 0; j < n; ++j) if (g.dangling[j]) D += s[j];
        res = 0.0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : res) if (n_threads > 1)
        for (uint32_t i = 0; i < n; ++i) {
            double acc = 0.0;
            if (member[i]) {
                for (uint64_t p = g.rowptr[i]; p < g.rowptr[i + 1]; ++p) acc += g.w_f64[p] * s[g.src[p]];
                acc += (D - (g.dangling[i] ? s[i] : 0.0)) * inv_nm1;
                double pi = pretrust ? pretrust[i] : initial;
                acc = (1.0 - alpha) * acc + alpha * pi;
            }
            ns[i] = acc;
            res += acc > s[i] ? acc - s[i] : s[i] - acc;
        }
        double *tmp = s; s = ns; ns = tmp;
    }
    if (loop_seconds) *loop_seconds = now_s() - t0;
    memcpy(out_f64, s, (size_t)n * sizeof(double));
    if (residual_l1) *residual_l1 = res;
    if (out_ld) {
        long double *a = (long double *)calloc(n, sizeof(long double)), *b = (long double *)calloc(n, sizeof(long double));
        for (uint32_t i = 0; i < n; ++i) a[i] = member[i] ? (long double)initial : 0.0L;
        long double inv_l = 1.0L / (long double)(g.n_valid - 1);
        for (uint32_t it = 0; it < num_iter; ++it) {
            long double D = 0.0L;
            for (uint32_t j = 0; j < n; ++j) if (g.dangling[j]) D += a[j];
            /* rows are independent and each is summed in list order: threads do not change a single bit */
#pragma omp parallel for schedule(dynamic, 1024) if (n_threads > 1)
            for (uint32_t i = 0; i < n; ++i) {
                long double acc = 0.0L;
                if (member[i]) {
                    for (uint64_t p = g.rowptr[i]; p < g.rowptr[i + 1]; ++p) acc += g.w_ld[p] * a[g.src[p]];
                    acc += (D - (g.dangling[i] ? a[i] : 0.0L)) * inv_l;
                    long double pi = pretrust ? (long double)pretrust[i] : (long double)initial;
                    acc = (1.0L - (long double)alpha) * acc + (long double)alpha * pi;
                }
                b[i] = acc;
            }
            long double *tmp = a; a = b; b = tmp;
        }
        memcpy(out_ld, a, (size_t)n * sizeof(long double));
        free(a); free(b);
    }
    free(s); free(ns); sg_free(&g);
    return 0;
}

int oracle_long_double_bytes(void) { return (int)sizeof(long double); }
int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
