/*
 * jess_oracle.c -- CPU restatement of the template-matching path that enzymm
 * delegates to PyJess/Jess.   *** TEST INFRASTRUCTURE ONLY ***
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.  The product (enzymm_amd/) never does.
 *
 * What it restates
 * ----------------
 * The call `pyjess.Jess(templates).query(molecule, rmsd_threshold,
 * distance_cutoff, max_dynamic_distance, max_candidates=..., best_match=...,
 * ignore_chain=...)` made at /root/reference/enzymm/jess_run.py:800-811 and in
 * /root/reference/tests/test_jess_run.py:32-40,54-57.
 *
 * The arithmetic behind that call lives in the third-party dependency
 * `pyjess ~=0.5.0` (/root/reference/pyproject.toml:30), a Cython wrapper of
 * the C program Jess (Barker & Thornton, Bioinformatics 19:1644, 2003).  Its
 * source is NOT in /root/reference and cannot be fetched, so this file
 * restates the published algorithm -- exhaustive constraint search over
 * template-atom -> target-atom assignments followed by least-squares
 * superposition -- and is pinned against every known answer the reference's
 * own tests hold (tests/golden/golden.json: G1, G2, A1, A2, A3; see
 * tests/test_oracle_golden.py).  Semantics those vectors do NOT pin are
 * marked [UNPINNED] below and in DESIGN.md.
 *
 * Semantics ("hypothesis E")
 * ------------------------------------------
 *  - target atoms: every ATOM/HETATM record up to the first ENDMDL.
 *  - residue admissibility: target residue name must be one of the template
 *    atom's residue names (3-letter name of cols 18-20 plus the expansion of
 *    the one-letter list in cols 55-60); skipped when match_mode >= 100.
 *  - atom-name admissibility by (match_mode % 100), compared on the raw
 *    4-character PDB name field:
 *        0      exact name
 *        1, 3   same element columns (first two characters of the field)
 *               [mode 1 UNPINNED: no reference test touches it]
 *        8      exact name, or the histidine ring-flip partner
 *               (ND1<->CD2, NE2<->CE1)
 *    any other mode is rejected with ORACLE_EUNSUPPORTED, never guessed.
 *  - template atoms sharing (chain, resSeq) form one template residue; they
 *    must map into ONE target residue (chain, resSeq, iCode); different
 *    template residues map to different target residues; no target atom is
 *    used twice.
 *  - every pair (i, j) of template atoms satisfies
 *        | |q_i - q_j| - |t_i - t_j| | <= thr_ij,
 *        thr_ij = min(distance_cutoff + w_i + w_j, max_dynamic_distance)
 *    in IEEE double.  enzymm always passes max_dynamic_distance ==
 *    distance_cutoff (jess_run.py:564-571), where thr_ij == distance_cutoff;
 *    the weighted form is [UNPINNED].
 *  - every complete assignment ("candidate") is superposed onto the template
 *    with the optimal PROPER rotation; RMSD = sqrt(sum |R q + t - p|^2 / k).
 *  - a candidate is a hit when RMSD <= rmsd_threshold.  best_match keeps, per
 *    template, the hit of minimum RMSD.  Hits whose RMSD lies within
 *    ORACLE_TIE_EPS (1e-9 A) of that GLOBAL minimum (exact symmetric copies of
 *    a site, e.g. the chains of a homo-oligomer built by symmetry operators)
 *    count as tied with it and the first of them in enumeration order -- the
 *    lexicographically smallest tuple of target atom indices -- is kept.  The
 *    rule does not depend on the order in which candidates are met (a running
 *    "better than the best so far by more than eps" comparison would).
 *    [UNPINNED: which of two tied sites PyJess reports depends on Jess's
 *    traversal order.]
 *  - max_candidates: at most that many candidates are evaluated per
 *    (template, molecule); if more exist the search stops and `truncated` is
 *    set.  Which candidates Jess would have seen first is [UNPINNED]; parity
 *    is only defined for searches that are not truncated.
 *  - ignore_chain == 0: two template atoms are on the same chain iff their
 *    targets are [UNPINNED: enzymm always passes ignore_chain=True].
 *
 * Enumeration order: template atoms in file order, target atoms in file
 * order -- so hits come out in lexicographic order of their index tuples.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_MAXK 64
#define ORACLE_EUNSUPPORTED (-2)
#define ORACLE_EINVAL (-1)
#define ORACLE_TIE_EPS 1e-9

/* ------------------------------------------------------------------ names */

static int element_eq(const char *a, const char *b) { return a[0] == b[0] && a[1] == b[1]; }

static const char *his_flip_partner(const char *name)
{
    if (!memcmp(name, " ND1", 4)) return " CD2";
    if (!memcmp(name, " CD2", 4)) return " ND1";
    if (!memcmp(name, " NE2", 4)) return " CE1";
    if (!memcmp(name, " CE1", 4)) return " NE2";
    return NULL;
}

/* 1 admissible, 0 not, <0 unsupported mode */
static int atom_name_admissible(int mode, const char *tname, const char *qname)
{
    switch (mode % 100) {
    case 0:
        return memcmp(tname, qname, 4) == 0;
    case 1: /* [UNPINNED] treated like 3 */
    case 3:
        return element_eq(tname, qname);
    case 8: {
        const char *p;
        if (memcmp(tname, qname, 4) == 0) return 1;
        p = his_flip_partner(tname);
        return p != NULL && memcmp(p, qname, 4) == 0;
    }
    default:
        return ORACLE_EUNSUPPORTED;
    }
}

static int residue_admissible(int mode, const char *tres, int ntres, const char *qres)
{
    int r;
    if (mode >= 100) return 1;
    for (r = 0; r < ntres; ++r)
        if (memcmp(tres + 3 * r, qres, 3) == 0) return 1;
    return 0;
}

/* --------------------------------------------------------- superposition */

static void jacobi3(double a[3][3], double v[3][3], double w[3])
{
    /* cyclic Jacobi eigen-decomposition of a symmetric 3x3 matrix */
    int i, j, sweep, p, q;
    for (i = 0; i < 3; ++i)
        for (j = 0; j < 3; ++j) v[i][j] = (i == j);
    for (sweep = 0; sweep < 64; ++sweep) {
        double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        if (off < 1e-300) break;
        for (p = 0; p < 2; ++p)
            for (q = p + 1; q < 3; ++q) {
                double apq = a[p][q], theta, t, c, s;
                if (fabs(apq) < 1e-300) continue;
                theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                c = 1.0 / sqrt(t * t + 1.0);
                s = t * c;
                for (i = 0; i < 3; ++i) { /* A <- A J */
                    double aip = a[i][p], aiq = a[i][q];
                    a[i][p] = c * aip - s * aiq;
                    a[i][q] = s * aip + c * aiq;
                }
                for (i = 0; i < 3; ++i) { /* A <- J^T A */
                    double api = a[p][i], aqi = a[q][i];
                    a[p][i] = c * api - s * aqi;
                    a[q][i] = s * api + c * aqi;
                }
                for (i = 0; i < 3; ++i) {
                    double vip = v[i][p], viq = v[i][q];
                    v[i][p] = c * vip - s * viq;
                    v[i][q] = s * vip + c * viq;
                }
            }
    }
    for (i = 0; i < 3; ++i) w[i] = a[i][i];
}

static void cross3(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

static double norm3(const double a[3]) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

/* any unit vector orthogonal to a (a is unit) */
static void any_orthogonal(const double a[3], double o[3])
{
    double e[3] = {0, 0, 0}, n;
    int m = 0;
    if (fabs(a[1]) < fabs(a[m])) m = 1;
    if (fabs(a[2]) < fabs(a[m])) m = 2;
    e[m] = 1.0;
    cross3(a, e, o);
    n = norm3(o);
    o[0] /= n; o[1] /= n; o[2] /= n;
}

/*
 * Kabsch (1978) superposition of q (target atoms) onto p (template atoms):
 * finds proper rotation R and translation t minimising sum |R q_i + t - p_i|^2.
 * rot is row-major.  Returns the RMSD, evaluated directly from the
 * transformed coordinates.
 */
static double superpose(int k, const double (*q)[3], const double (*p)[3], double rot[9], double trans[3])
{
    double qc[3] = {0, 0, 0}, pc[3] = {0, 0, 0};
    double H[3][3] = {{0}}, HtH[3][3], V[3][3], w[3];
    double v[3][3], u[3][3], sum;
    int i, a, b, order[3] = {0, 1, 2};

    for (i = 0; i < k; ++i)
        for (a = 0; a < 3; ++a) { qc[a] += q[i][a]; pc[a] += p[i][a]; }
    for (a = 0; a < 3; ++a) { qc[a] /= k; pc[a] /= k; }
    /* H[a][b] = sum (p_a)(q_b): R = argmax tr(R^T H) maps q -> p */
    for (i = 0; i < k; ++i)
        for (a = 0; a < 3; ++a)
            for (b = 0; b < 3; ++b) H[a][b] += (p[i][a] - pc[a]) * (q[i][b] - qc[b]);
    for (a = 0; a < 3; ++a)
        for (b = 0; b < 3; ++b)
            HtH[a][b] = H[0][a] * H[0][b] + H[1][a] * H[1][b] + H[2][a] * H[2][b];
    jacobi3(HtH, V, w);
    /* sort eigenvalues descending */
    for (a = 0; a < 2; ++a)
        for (b = a + 1; b < 3; ++b)
            if (w[order[b]] > w[order[a]]) { int tmp = order[a]; order[a] = order[b]; order[b] = tmp; }
    for (a = 0; a < 2; ++a)
        for (b = 0; b < 3; ++b) v[a][b] = V[b][order[a]];
    cross3(v[0], v[1], v[2]); /* right-handed triad */
    /* u_a = H v_a / |H v_a| for the two dominant directions */
    for (a = 0; a < 2; ++a) {
        double n;
        for (b = 0; b < 3; ++b) u[a][b] = H[b][0] * v[a][0] + H[b][1] * v[a][1] + H[b][2] * v[a][2];
        if (a == 1) { /* Gram-Schmidt against u0 */
            double d = u[1][0] * u[0][0] + u[1][1] * u[0][1] + u[1][2] * u[0][2];
            for (b = 0; b < 3; ++b) u[1][b] -= d * u[0][b];
        }
        n = norm3(u[a]);
        if (n < 1e-12) { /* rank-deficient covariance: any orthogonal completion is optimal */
            if (a == 0) { u[0][0] = 1; u[0][1] = 0; u[0][2] = 0; }
            else any_orthogonal(u[0], u[1]);
        } else
            for (b = 0; b < 3; ++b) u[a][b] /= n;
    }
    cross3(u[0], u[1], u[2]);
    /* R = sum_a u_a v_a^T */
    for (a = 0; a < 3; ++a)
        for (b = 0; b < 3; ++b)
            rot[3 * a + b] = u[0][a] * v[0][b] + u[1][a] * v[1][b] + u[2][a] * v[2][b];
    for (a = 0; a < 3; ++a)
        trans[a] = pc[a] - (rot[3 * a] * qc[0] + rot[3 * a + 1] * qc[1] + rot[3 * a + 2] * qc[2]);
    sum = 0.0;
    for (i = 0; i < k; ++i)
        for (a = 0; a < 3; ++a) {
            double x = rot[3 * a] * q[i][0] + rot[3 * a + 1] * q[i][1] + rot[3 * a + 2] * q[i][2] + trans[a] - p[i][a];
            sum += x * x;
        }
    return sqrt(sum / k);
}

/* ------------------------------------------------------------------ search */

typedef struct {
    int n, k;
    const double (*qxyz)[3];
    const double (*txyz)[3];
    const int *qresid;  /* target residue ordinal per atom */
    const char *qchain; /* n*2 */
    const char *tchain; /* k*2 */
    int tgroup[ORACLE_MAXK];
    double D[ORACLE_MAXK][ORACLE_MAXK];
    double thr[ORACLE_MAXK][ORACLE_MAXK];
    int *cand[ORACLE_MAXK];
    int ncand[ORACLE_MAXK];
    int assign[ORACLE_MAXK];
    double rmsd_thr;
    int max_candidates, best_match, ignore_chain;
    /* results */
    int max_hits, nhits, candidates, truncated, stop;
    int *out_atoms;
    double *out_rmsd, *out_rot, *out_trans;
    /* best_match: the hits within ORACLE_TIE_EPS of the minimum RMSD met so far, in enumeration order */
    struct contender { double rmsd, rot[9], trans[3]; int atoms[ORACLE_MAXK]; } *tied;
    int ntied, tied_cap, failed;
    double min_rmsd;
} search_t;

static double dist3(const double a[3], const double b[3])
{
    double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return sqrt(dx * dx + dy * dy + dz * dz);
}

static void on_candidate(search_t *s)
{
    double q[ORACLE_MAXK][3], rot[9], trans[3], rmsd;
    int i, slot;
    if (s->candidates >= s->max_candidates) { s->truncated = 1; s->stop = 1; return; }
    s->candidates++;
    for (i = 0; i < s->k; ++i) memcpy(q[i], s->qxyz[s->assign[i]], sizeof q[i]);
    rmsd = superpose(s->k, (const double (*)[3])q, s->txyz, rot, trans);
    if (!(rmsd <= s->rmsd_thr)) return;
    if (s->best_match) {
        /* keep every hit within eps of the running minimum; a new minimum drops those it leaves behind.  At the
         * end the list holds exactly the hits within eps of the global minimum, earliest first (query_one) */
        int w = 0;
        if (s->ntied == 0 || rmsd < s->min_rmsd) s->min_rmsd = rmsd;
        for (i = 0; i < s->ntied; ++i)
            if (s->tied[i].rmsd <= s->min_rmsd + ORACLE_TIE_EPS) { if (w != i) s->tied[w] = s->tied[i]; ++w; }
        s->ntied = w;
        if (!(rmsd <= s->min_rmsd + ORACLE_TIE_EPS)) return;
        if (s->ntied == s->tied_cap) {
            int cap = s->tied_cap ? 2 * s->tied_cap : 4;
            struct contender *grown = realloc(s->tied, sizeof *grown * (size_t)cap);
            if (!grown) { s->stop = 1; s->failed = 1; return; }
            s->tied = grown; s->tied_cap = cap;
        }
        s->tied[s->ntied].rmsd = rmsd;
        memcpy(s->tied[s->ntied].rot, rot, sizeof rot);
        memcpy(s->tied[s->ntied].trans, trans, sizeof trans);
        for (i = 0; i < s->k; ++i) s->tied[s->ntied].atoms[i] = s->assign[i];
        ++s->ntied;
        return;
    } else {
        if (s->nhits >= s->max_hits) { s->truncated = 1; s->stop = 1; return; }
        slot = s->nhits++;
    }
    for (i = 0; i < s->k; ++i) s->out_atoms[slot * s->k + i] = s->assign[i];
    s->out_rmsd[slot] = rmsd;
    memcpy(s->out_rot + 9 * slot, rot, sizeof rot);
    memcpy(s->out_trans + 3 * slot, trans, sizeof trans);
}

static void descend(search_t *s, int level)
{
    int ci, j;
    for (ci = 0; ci < s->ncand[level] && !s->stop; ++ci) {
        int c = s->cand[level][ci], ok = 1;
        for (j = 0; j < level && ok; ++j) {
            int a = s->assign[j];
            int same_t, same_q;
            if (a == c) { ok = 0; break; }
            same_t = s->tgroup[level] == s->tgroup[j];
            same_q = s->qresid[c] == s->qresid[a];
            if (same_t != same_q) { ok = 0; break; }
            if (!s->ignore_chain) {
                int tc = memcmp(s->tchain + 2 * level, s->tchain + 2 * j, 2) == 0;
                int qc = memcmp(s->qchain + 2 * c, s->qchain + 2 * a, 2) == 0;
                if (tc != qc) { ok = 0; break; }
            }
            if (fabs(dist3(s->qxyz[c], s->qxyz[a]) - s->D[level][j]) > s->thr[level][j]) ok = 0;
        }
        if (!ok) continue;
        s->assign[level] = c;
        if (level == s->k - 1) on_candidate(s);
        else descend(s, level + 1);
    }
}

/* target residue ordinals from (chain, resSeq, iCode); O(n log n) */
typedef struct { char chain[2]; char icode; int resseq; int idx; } reskey_t;

static int reskey_cmp(const void *a, const void *b)
{
    const reskey_t *x = a, *y = b;
    int c = memcmp(x->chain, y->chain, 2);
    if (c) return c;
    if (x->resseq != y->resseq) return x->resseq < y->resseq ? -1 : 1;
    if (x->icode != y->icode) return x->icode < y->icode ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

int oracle_residue_ids(int n, const char *qchain, const int *qresseq, const char *qicode, int *out)
{
    reskey_t *keys = malloc(sizeof(reskey_t) * (n > 0 ? n : 1));
    int i, id = -1;
    if (!keys) return ORACLE_EINVAL;
    for (i = 0; i < n; ++i) {
        memcpy(keys[i].chain, qchain + 2 * i, 2);
        keys[i].icode = qicode[i];
        keys[i].resseq = qresseq[i];
        keys[i].idx = i;
    }
    qsort(keys, n, sizeof(reskey_t), reskey_cmp);
    for (i = 0; i < n; ++i) {
        if (i == 0 || memcmp(keys[i].chain, keys[i - 1].chain, 2) || keys[i].resseq != keys[i - 1].resseq ||
            keys[i].icode != keys[i - 1].icode)
            ++id;
        out[keys[i].idx] = id;
    }
    free(keys);
    return id + 1;
}

/*
 * One template against one molecule.  Returns the number of hits written
 * (best_match: 0 or 1), or a negative error code.
 */
/* scratch != NULL: room for ORACLE_MAXK * max(n, 1) ints, owned by the caller (oracle_query_library keeps one per
 * thread: 15 allocations of n ints per template, freed again, make glibc trim and re-fault its heaps, which
 * serialises many threads on the kernel's address-space lock) */
static int query_one(
    int n, const double *qxyz, const char *qname, const char *qres, const char *qchain, const int *qresid,
    int k, const double *txyz, const int *tmode, const char *tname, const char *tresnames, const int *tnres,
    int maxresn, const char *tchain, const int *tresseq, const double *tweight,
    double rmsd_thr, double dist_cutoff, double max_dyn, int max_candidates, int best_match, int ignore_chain,
    int max_hits, int *out_atoms, double *out_rmsd, double *out_rot, double *out_trans, int *out_candidates,
    int *out_truncated, int *scratch);

int oracle_query(
    /* molecule */
    int n, const double *qxyz, const char *qname, const char *qres, const char *qchain, const int *qresid,
    /* template */
    int k, const double *txyz, const int *tmode, const char *tname, const char *tresnames, const int *tnres,
    int maxresn, const char *tchain, const int *tresseq, const double *tweight,
    /* parameters */
    double rmsd_thr, double dist_cutoff, double max_dyn, int max_candidates, int best_match, int ignore_chain,
    /* results */
    int max_hits, int *out_atoms, double *out_rmsd, double *out_rot, double *out_trans, int *out_candidates,
    int *out_truncated)
{
    return query_one(n, qxyz, qname, qres, qchain, qresid, k, txyz, tmode, tname, tresnames, tnres, maxresn, tchain,
                     tresseq, tweight, rmsd_thr, dist_cutoff, max_dyn, max_candidates, best_match, ignore_chain,
                     max_hits, out_atoms, out_rmsd, out_rot, out_trans, out_candidates, out_truncated, NULL);
}

static int query_one(
    int n, const double *qxyz, const char *qname, const char *qres, const char *qchain, const int *qresid,
    int k, const double *txyz, const int *tmode, const char *tname, const char *tresnames, const int *tnres,
    int maxresn, const char *tchain, const int *tresseq, const double *tweight,
    double rmsd_thr, double dist_cutoff, double max_dyn, int max_candidates, int best_match, int ignore_chain,
    int max_hits, int *out_atoms, double *out_rmsd, double *out_rot, double *out_trans, int *out_candidates,
    int *out_truncated, int *scratch)
{
    search_t s;
    int i, j, a, ngroups = 0, rc = 0;
    if (k <= 0 || k > ORACLE_MAXK || n < 0 || max_hits < 1) return ORACLE_EINVAL;
    memset(&s, 0, sizeof s);
    s.n = n; s.k = k;
    s.qxyz = (const double (*)[3])qxyz;
    s.txyz = (const double (*)[3])txyz;
    s.qresid = qresid; s.qchain = qchain; s.tchain = tchain;
    s.rmsd_thr = rmsd_thr; s.max_candidates = max_candidates;
    s.best_match = best_match; s.ignore_chain = ignore_chain;
    s.max_hits = max_hits; s.out_atoms = out_atoms; s.out_rmsd = out_rmsd;
    s.out_rot = out_rot; s.out_trans = out_trans;
    /* template residues: atoms sharing (chain, resSeq) */
    for (i = 0; i < k; ++i) {
        s.tgroup[i] = -1;
        for (j = 0; j < i; ++j)
            if (tresseq[i] == tresseq[j] && memcmp(tchain + 2 * i, tchain + 2 * j, 2) == 0) { s.tgroup[i] = s.tgroup[j]; break; }
        if (s.tgroup[i] < 0) s.tgroup[i] = ngroups++;
    }
    for (i = 0; i < k; ++i)
        for (j = 0; j < k; ++j) {
            double t = dist_cutoff + tweight[i] + tweight[j];
            s.D[i][j] = dist3(s.txyz[i], s.txyz[j]);
            s.thr[i][j] = t < max_dyn ? t : max_dyn;
        }
    /* admissible target atoms per template atom */
    for (i = 0; i < k; ++i) {
        s.cand[i] = scratch ? scratch + (size_t)i * (n > 0 ? n : 1) : malloc(sizeof(int) * (n > 0 ? n : 1));
        if (!s.cand[i]) { rc = ORACLE_EINVAL; goto done; }
        for (a = 0; a < n; ++a) {
            int ok;
            if (!residue_admissible(tmode[i], tresnames + 3 * maxresn * i, tnres[i], qres + 3 * a)) continue;
            ok = atom_name_admissible(tmode[i], tname + 4 * i, qname + 4 * a);
            if (ok < 0) { rc = ok; goto done; }
            if (ok) s.cand[i][s.ncand[i]++] = a;
        }
    }
    descend(&s, 0);
    if (s.failed) { rc = ORACLE_EINVAL; goto done; }
    if (best_match && s.ntied > 0) { /* the earliest of the hits tied with the global minimum */
        for (i = 0; i < k; ++i) out_atoms[i] = s.tied[0].atoms[i];
        out_rmsd[0] = s.tied[0].rmsd;
        memcpy(out_rot, s.tied[0].rot, sizeof s.tied[0].rot);
        memcpy(out_trans, s.tied[0].trans, sizeof s.tied[0].trans);
        s.nhits = 1;
    }
    rc = s.nhits;
    if (out_candidates) *out_candidates = s.candidates;
    if (out_truncated) *out_truncated = s.truncated;
done:
    free(s.tied);
    if (!scratch)
        for (i = 0; i < k; ++i) free(s.cand[i]);
    return rc;
}

/*
 * Many templates (concatenated, toff[nt+1] atom offsets) against one
 * molecule, best_match semantics, per-template parameters.  out_* hold one
 * slot per template; out_hit[t] = 1 when template t produced a hit.
 * Templates are independent, so the loop is OpenMP-parallel when built with
 * -fopenmp (bench.py's cpu_baseline states the thread count it used).
 */
int oracle_query_library(
    int n, const double *qxyz, const char *qname, const char *qres, const char *qchain, const int *qresid,
    int nt, const int *toff, const double *txyz, const int *tmode, const char *tname, const char *tresnames,
    const int *tnres, int maxresn, const char *tchain, const int *tresseq, const double *tweight,
    const double *rmsd_thr, const double *dist_cutoff, const double *max_dyn, int max_candidates, int ignore_chain,
    int maxk, int *out_hit, int *out_atoms /* nt*maxk */, double *out_rmsd, double *out_rot, double *out_trans,
    int *out_candidates, int *out_truncated)
{
    int t, err = 0;
#pragma omp parallel
    {
        int *scratch = malloc(sizeof(int) * (size_t)ORACLE_MAXK * (n > 0 ? n : 1));  /* one per thread */
#pragma omp for schedule(dynamic, 8)
        for (t = 0; t < nt; ++t) {
            int o = toff[t], k = toff[t + 1] - o, rc;
            int atoms[ORACLE_MAXK];
            if (!scratch) {
#pragma omp critical
                err = ORACLE_EINVAL;
                out_hit[t] = 0;
                continue;
            }
            rc = query_one(n, qxyz, qname, qres, qchain, qresid, k, txyz + 3 * o, tmode + o, tname + 4 * o,
                           tresnames + 3 * maxresn * o, tnres + o, maxresn, tchain + 2 * o, tresseq + o, tweight + o,
                           rmsd_thr[t], dist_cutoff[t], max_dyn[t], max_candidates, 1, ignore_chain, 1, atoms,
                           out_rmsd + t, out_rot + 9 * t, out_trans + 3 * t, out_candidates + t, out_truncated + t,
                           scratch);
            if (rc < 0) {
#pragma omp critical
                err = rc;
                out_hit[t] = 0;
                continue;
            }
            out_hit[t] = rc;
            if (rc > 0) memcpy(out_atoms + (size_t)maxk * t, atoms, sizeof(int) * k);
        }
        free(scratch);
    }
    return err;
}
