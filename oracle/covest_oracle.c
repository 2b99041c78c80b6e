/*
 * covest_oracle.c -- CPU restatement of CovEst's likelihood hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the timed CPU
 * baseline.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.  Nothing under covest_amd/ (the product) may import, link or
 * call it: the product path is the HIP library and fails loudly without it.
 *
 * Parity status: PINNED.  The reference's own tests assert no likelihood value
 * (reference tests/test_models.py:8-19 only checks the model registry), so the
 * oracle is pinned by golden vectors generated in the build container by
 * importing the reference itself (tests/golden/make_golden.py, fixtures under
 * tests/golden/ *.json) and, when oracle/_ref/ is built, by calling the
 * reference's compiled C extension side by side (tests/test_oracle_vs_ref.py).
 *
 * Every function cites the reference file:line it restates.  Paths are relative
 * to the reference checkout (mhozza/covest v0.5.6).
 *
 * Arithmetic notes that define "faithful":
 *   - x86-64 `long double` (80-bit x87) for the pmf product, as the reference.
 *   - Python 3.10 builtin sum() == naive left-to-right double adds.
 *   - math.fsum == exactly rounded sum (Shewchuk partials), restated below.
 *   - float ** int and int ** -int in Python end in libm pow().
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_MAX_PARAMS 5
#define ORACLE_CHUNK_LEN 200.0L /* reference c_src/covest_poissonmodule.c:5 (MAX_EXP) */

typedef struct {
    int kind;         /* 0 = basic (2 params), 1 = repeats (5 params) */
    int k;            /* k-mer size */
    int r;            /* read length */
    int n_err;        /* S = max_error: number of error classes summed */
    const double *comb; /* [n_err] C(k,s)*3^s as the reference's self.comb */
    int64_t n_keys;
    const int32_t *keys;   /* histogram keys j, in dict order */
    const double *counts;  /* h_j as doubles (exact for h < 2^53) */
    double tail;
    double lo[ORACLE_MAX_PARAMS]; /* NaN = None */
    double hi[ORACLE_MAX_PARAMS];
    double threshold;   /* repeats: b_o cut-off */
    int has_threshold;  /* 0 = threshold None */
} oracle_model;

/* --- covest_poisson.truncated_poisson: c_src/covest_poissonmodule.c:7-35 ---
 * TP(l, j) = l^j / (j! * (e^l - 1)).  The product of the j factors l/i is kept
 * in long double, each factor rounded to double first (:22-24); then e^l is
 * divided out 200 at a time (:25-28) and the residual e^l - 1 (:29-31).
 * l == 0 or NaN: the reference returns Py_BuildValue("d", 0), passing an int
 * for a double vararg (undefined behaviour, :15-17); its intent, 0.0, is used
 * here.  Quirk (iv) of SURVEY 8(a)/A1 (residual l <= 1e-8 keeps p3 = the
 * ORIGINAL l) is reproduced as written because it costs nothing. */
double oracle_truncated_poisson(double l, int j)
{
    if (l == 0 || l != l)
        return 0.0;
    long double prod = 1.0L;
    long double denom = l;
    for (int i = 1; i <= j; i++) {
        double factor = l / i;
        prod *= factor;
    }
    while (l > 200 && prod > 0) {
        prod /= expl(ORACLE_CHUNK_LEN);
        l -= 200;
    }
    if (l > 1e-8 && prod > 0)
        denom = expl(l) - 1;
    return (double)(prod / denom);
}

/* math.fsum: exactly rounded sum of doubles (CPython Modules/mathmodule.c,
 * Shewchuk's algorithm).  Used at covest/models.py:103.  Special values: any
 * NaN input -> NaN; +inf and -inf together -> NaN (CPython raises ValueError,
 * which cannot happen here: p_j >= 0); otherwise inf propagates. */
static double exact_sum(const double *x, int64_t n)
{
    int64_t cap = 32, m = 0;
    double *part = (double *)malloc(sizeof(double) * cap);
    double special = 0.0;
    int have_special = 0;
    for (int64_t t = 0; t < n; t++) {
        double v = x[t];
        if (!isfinite(v)) {
            special += v;
            have_special = 1;
            continue;
        }
        int64_t keep = 0;
        for (int64_t u = 0; u < m; u++) {
            double y = part[u];
            if (fabs(v) < fabs(y)) {
                double tmp = v;
                v = y;
                y = tmp;
            }
            volatile double hi = v + y;
            volatile double yr = hi - v;
            double lo = y - yr;
            if (lo != 0.0)
                part[keep++] = lo;
            v = hi;
        }
        m = keep;
        if (v != 0.0) {
            if (m == cap) {
                cap *= 2;
                part = (double *)realloc(part, sizeof(double) * cap);
            }
            part[m++] = v;
        }
    }
    double total = 0.0;
    if (have_special) {
        free(part);
        return special;
    }
    if (m > 0) {
        int64_t u = m;
        total = part[--u];
        double lo = 0.0;
        while (u > 0) {
            double v = total;
            double y = part[--u];
            volatile double hi = v + y;
            volatile double yr = hi - v;
            lo = y - yr;
            total = hi;
            if (lo != 0.0)
                break;
        }
        /* round-half-even correction, as CPython does */
        if (u > 0 && ((lo < 0.0 && part[u - 1] < 0.0) || (lo > 0.0 && part[u - 1] > 0.0))) {
            double y = lo * 2.0;
            volatile double v = total + y;
            double yr = v - total;
            if (y == yr)
                total = v;
        }
    }
    free(part);
    return total;
}

/* utils.safe_log: covest/utils.py:32-35 (x None or <= 0 -> -inf). */
static double safe_log(double x)
{
    if (x <= 0)
        return -INFINITY;
    return log(x);
}

/* BasicModel.fit_to_bounds: covest/models.py:60-69. None bounds are NaN here. */
static void clamp_to_bounds(const oracle_model *m, int n_par, const double *in, double *out)
{
    for (int i = 0; i < n_par; i++) {
        double v = in[i];
        double lo = m->lo[i], hi = m->hi[i];
        if (lo == lo && v < lo)
            v = lo;
        else if (hi == hi && v > hi)
            v = hi;
        out[i] = v;
    }
}

/* correct_c + _get_lambda_s: covest/models.py:71-79.
 * ck = c*(r-k+1)/r ; l_s = ((ck * 3**-s) * (1-e)**(k-s)) * e**s  */
static void error_class_rates(const oracle_model *m, double c, double err, double *l_s)
{
    double ck = c * (double)(m->r - m->k + 1) / (double)m->r;
    for (int s = 0; s < m->n_err; s++) {
        double v = ck * pow(3.0, (double)-s);
        v = v * pow(1.0 - err, (double)(m->k - s));
        v = v * pow(err, (double)s);
        l_s[s] = v;
    }
}

/* BasicModel.compute_probabilities: covest/models.py:81-98. */
static void basic_probabilities(const oracle_model *m, const double *par, double *p_out)
{
    const int S = m->n_err;
    double l_s[64], n_s[64], a_s[64];
    error_class_rates(m, par[0], par[1], l_s);
    double tot = 0.0;
    for (int s = 0; s < S; s++) {
        n_s[s] = m->comb[s] * (1.0 - exp(-l_s[s]));
        tot += n_s[s];
    }
    if (tot == 0) /* utils.fix_zero, covest/utils.py:25-29 */
        tot = 1;
    for (int s = 0; s < S; s++)
        a_s[s] = n_s[s] / tot;
    for (int64_t b = 0; b < m->n_keys; b++) {
        int j = m->keys[b];
        double acc = 0.0;
        for (int s = 0; s < S; s++)
            acc += a_s[s] * oracle_truncated_poisson(l_s[s], j);
        p_out[b] = acc;
    }
}

/* RepeatsModel.get_b_o: covest/models.py:193-208. */
static double copy_number_weight(double q1, double q2, double q, int o)
{
    if (o == 0)
        return 0;
    if (o == 1)
        return q1;
    if (o == 2)
        return (1 - q1) * q2;
    return (1 - q1) * (1 - q2) * q * pow(1 - q, (double)(o - 3));
}

/* RepeatsModel.get_hist_threshold: covest/models.py:185-191.
 * First o in 1..max(hist)-1 with b_o(o) <= threshold, else max(hist). */
int oracle_threshold_o(double q1, double q2, double q, double threshold, int has_threshold,
                       int hist_max)
{
    if (has_threshold) {
        for (int o = 1; o < hist_max; o++)
            if (copy_number_weight(q1, q2, q, o) <= threshold)
                return o;
    }
    return hist_max;
}

static int largest_key(const oracle_model *m)
{
    int best = m->keys[0];
    for (int64_t b = 1; b < m->n_keys; b++)
        if (m->keys[b] > best)
            best = m->keys[b];
    return best;
}

/* RepeatsModel.compute_probabilities: covest/models.py:211-242. */
static void repeats_probabilities(const oracle_model *m, const double *par, double *p_out)
{
    const int S = m->n_err;
    const double q1 = par[2], q2 = par[3], q = par[4];
    const int T = oracle_threshold_o(q1, q2, q, m->threshold, m->has_threshold, largest_key(m));
    double l_s[64];
    error_class_rates(m, par[0], par[1], l_s);
    const int n_o = T > 1 ? T - 1 : 0;
    double *a_os = (double *)malloc(sizeof(double) * (size_t)(n_o > 0 ? n_o : 1) * S);
    double *b_o = (double *)malloc(sizeof(double) * (size_t)(n_o > 0 ? n_o : 1));
    for (int o = 1; o < T; o++) {
        double *row = a_os + (size_t)(o - 1) * S;
        double tot = 0.0;
        for (int s = 0; s < S; s++) {
            row[s] = m->comb[s] * (1.0 - exp(o * -l_s[s]));
            tot += row[s];
        }
        if (tot == 0) /* fix_zero at :225, then the redundant test at :231 */
            tot = 1;
        for (int s = 0; s < S; s++)
            row[s] = row[s] / tot;
        b_o[o - 1] = copy_number_weight(q1, q2, q, o);
    }
    for (int64_t b = 0; b < m->n_keys; b++) {
        int j = m->keys[b];
        double outer = 0.0;
        for (int o = 1; o < T; o++) {
            const double *row = a_os + (size_t)(o - 1) * S;
            double inner = 0.0;
            for (int s = 0; s < S; s++)
                inner += row[s] * oracle_truncated_poisson(o * l_s[s], j);
            outer += b_o[o - 1] * inner;
        }
        p_out[b] = outer;
    }
    free(a_os);
    free(b_o);
}

int oracle_param_count(const oracle_model *m) { return m->kind == 0 ? 2 : 5; }

/* compute_probabilities after fit_to_bounds; p_out has n_keys entries. */
void oracle_probabilities(const oracle_model *m, const double *params, double *p_out)
{
    double par[ORACLE_MAX_PARAMS];
    clamp_to_bounds(m, oracle_param_count(m), params, par);
    if (m->kind == 0)
        basic_probabilities(m, par, p_out);
    else
        repeats_probabilities(m, par, p_out);
}

/* BasicModel.compute_loglikelihood: covest/models.py:100-107. */
double oracle_loglikelihood(const oracle_model *m, const double *params)
{
    double *p = (double *)malloc(sizeof(double) * (size_t)(m->n_keys > 0 ? m->n_keys : 1));
    oracle_probabilities(m, params, p);
    double sp = exact_sum(p, m->n_keys);
    if (!(sp < 1)) /* min(1, x): 1 unless x < 1 (NaN -> 1) */
        sp = 1;
    double tail_term = 0;
    if (sp < 1)
        tail_term = m->tail * safe_log(1 - sp);
    double acc = 0.0;
    for (int64_t b = 0; b < m->n_keys; b++) {
        double h = m->counts[b];
        if (h != 0)
            acc += h * safe_log(p[b]);
    }
    free(p);
    return acc + tail_term;
}

/* --- batch evaluation over host threads (compute_loglikelihood_multi,
 * covest/models.py:109-117, with threads instead of processes).  This is the
 * timed CPU baseline of bench.py. --- */
typedef struct {
    const oracle_model *m;
    const double *params;
    double *out;
    int64_t n;
    int n_par;
    int64_t next; /* shared work counter */
    pthread_mutex_t lock;
} batch_job;

static void *batch_worker(void *arg)
{
    batch_job *job = (batch_job *)arg;
    for (;;) {
        pthread_mutex_lock(&job->lock);
        int64_t i = job->next++;
        pthread_mutex_unlock(&job->lock);
        if (i >= job->n)
            break;
        job->out[i] = oracle_loglikelihood(job->m, job->params + i * job->n_par);
    }
    return NULL;
}

void oracle_loglikelihood_many(const oracle_model *m, int64_t n, const double *params, double *out,
                               int n_threads)
{
    batch_job job;
    job.m = m;
    job.params = params;
    job.out = out;
    job.n = n;
    job.n_par = oracle_param_count(m);
    job.next = 0;
    pthread_mutex_init(&job.lock, NULL);
    if (n_threads < 1)
        n_threads = 1;
    if (n_threads == 1) {
        batch_worker(&job);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
        for (int t = 0; t < n_threads; t++)
            pthread_create(&th[t], NULL, batch_worker, &job);
        for (int t = 0; t < n_threads; t++)
            pthread_join(th[t], NULL);
        free(th);
    }
    pthread_mutex_destroy(&job.lock);
}

/* ---------------------------------------------------------------------------
 * "Fast" CPU mode (CPU baseline row): the same likelihood with the
 * O(1) log-domain pmf term the GPU's general kernel uses instead of the reference's
 * O(j) long-double product, so that the reported GPU/CPU ratio can be split into
 * "better algorithm" and "faster machine".  It reproduces the reference's normaliser
 * (200-chunk division, x <= 1e-8 branches) like the kernels do; it is NOT bit-faithful
 * (lgamma rounding: <= 1e-11 relative per term) and is pinned to the faithful mode by
 * tests/test_oracle_golden.py::test_fast_mode_agrees.  Only zero-count-free work is
 * skipped when tail == 0, exactly as the HIP library does. */
static double log_norm_fast(double x, double log_x)
{
    if (x <= 1e-8)
        return log_x;
    double base = 0.0, xr = x;
    if (x > 200.0) {
        double n = ceil(x / 200.0) - 1.0;
        xr = x - 200.0 * n;
        if (xr > 200.0) {
            n += 1.0;
            xr -= 200.0;
        } else if (xr <= 0.0) {
            n -= 1.0;
            xr += 200.0;
        }
        base = 200.0 * n;
        if (xr <= 1e-8)
            return base + log_x;
    }
    if (xr < 1.0)
        return base + log(expm1(xr));
    return base + (xr + log1p(-exp(-xr)));
}

double oracle_loglikelihood_fast(const oracle_model *m, const double *params)
{
    double par[ORACLE_MAX_PARAMS];
    const int S = m->n_err;
    clamp_to_bounds(m, oracle_param_count(m), params, par);
    double l_s[64];
    error_class_rates(m, par[0], par[1], l_s);
    int T = 2;
    if (m->kind == 1)
        T = oracle_threshold_o(par[2], par[3], par[4], m->threshold, m->has_threshold, largest_key(m));
    const int n_o = T > 1 ? T - 1 : 0;
    double *lx = (double *)malloc(sizeof(double) * (size_t)(n_o > 0 ? n_o : 1) * S * 3);
    double *cc = lx + (size_t)(n_o > 0 ? n_o : 1) * S, *ww = cc + (size_t)(n_o > 0 ? n_o : 1) * S;
    for (int o = 1; o < T; o++) {
        double n_os[64], tot = 0.0;
        for (int s = 0; s < S; s++) {
            n_os[s] = m->comb[s] * (1.0 - exp(o * -l_s[s]));
            tot += n_os[s];
        }
        if (tot == 0)
            tot = 1;
        const double b = m->kind == 1 ? copy_number_weight(par[2], par[3], par[4], o) : 1.0;
        for (int s = 0; s < S; s++) {
            const size_t at = (size_t)(o - 1) * S + s;
            const double x = o * l_s[s];
            ww[at] = b * (n_os[s] / tot);
            if (x > 0) {
                lx[at] = log(x);
                cc[at] = -log_norm_fast(x, lx[at]);
            } else {
                lx[at] = 0.0;
                cc[at] = -INFINITY;
            }
        }
    }
    double acc = 0.0, sp = 0.0;
    for (int64_t b = 0; b < m->n_keys; b++) {
        const double h = m->counts[b];
        if (m->tail == 0 && h == 0)
            continue; /* the tail term is exactly 0: zero-count keys influence nothing */
        const int j = m->keys[b] > 0 ? m->keys[b] : 0;
        const double lg = lgamma((double)j + 1.0);
        double p = 0.0;
        for (size_t at = 0; at < (size_t)n_o * S; at++)
            if (ww[at] != 0)
                p += ww[at] * exp(j * lx[at] + cc[at] - lg);
        sp += p;
        if (h != 0)
            acc += h * safe_log(p);
    }
    free(lx);
    double tail_term = 0;
    if (!(sp < 1))
        sp = 1;
    if (m->tail != 0 && sp < 1)
        tail_term = m->tail * safe_log(1 - sp);
    return acc + tail_term;
}

static void *batch_worker_fast(void *arg)
{
    batch_job *job = (batch_job *)arg;
    for (;;) {
        pthread_mutex_lock(&job->lock);
        int64_t i = job->next++;
        pthread_mutex_unlock(&job->lock);
        if (i >= job->n)
            break;
        job->out[i] = oracle_loglikelihood_fast(job->m, job->params + i * job->n_par);
    }
    return NULL;
}

void oracle_loglikelihood_many_fast(const oracle_model *m, int64_t n, const double *params, double *out,
                                    int n_threads)
{
    batch_job job;
    job.m = m;
    job.params = params;
    job.out = out;
    job.n = n;
    job.n_par = oracle_param_count(m);
    job.next = 0;
    pthread_mutex_init(&job.lock, NULL);
    if (n_threads < 1)
        n_threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
    for (int t = 0; t < n_threads; t++)
        pthread_create(&th[t], NULL, batch_worker_fast, &job);
    for (int t = 0; t < n_threads; t++)
        pthread_join(th[t], NULL);
    free(th);
    pthread_mutex_destroy(&job.lock);
}

/* Sequential strict-< arg-min over a value list, as the scan in
 * covest/grid.py:65-70 with maximize=False: lowest index wins ties, NaN never
 * wins, +inf never beats the start value.  Returns -1 if nothing beats start. */
int64_t oracle_first_min(const double *vals, int64_t n, double start, double *best_out)
{
    int64_t arg = -1;
    double best = start;
    for (int64_t i = 0; i < n; i++) {
        if (vals[i] < best) {
            best = vals[i];
            arg = i;
        }
    }
    if (best_out)
        *best_out = best;
    return arg;
}

/* ---- histogram down-sampling: expected counts of covest/histogram.py:47-69 ----
 * out[j-1] += counts_s * probs_s[j-1] for every source bin s, probs as the reference builds them:
 * keys < 100: scipy's binomial pmf (here exp of long-double lgammal terms); keys >= 100: the C
 * extension's poisson_dist(i * prob, i), c_src/covest_poissonmodule.c:64-108, restated statement by
 * statement in long double when faithful != 0 -- INCLUDING its defect for i * prob > 200 (the rescaling
 * loop subtracts MAX_EXP from `l` itself) -- or the Poisson pmf (what the GPU path computes) when
 * faithful == 0.  Same O(i) work per source bin as the reference. */
void oracle_thin_expected(int64_t n, const int32_t *keys, const double *counts, double factor, int64_t out_len,
                          double *out, int faithful)
{
    const double prob = 1.0 / factor;
    for (int64_t j = 0; j < out_len; j++)
        out[j] = 0.0;
    for (int64_t s = 0; s < n; s++) {
        const int i = keys[s];
        const double v = counts[s];
        if (i < 100) {
            const long double lp = logl((long double)prob), lq = log1pl(-(long double)prob);
            for (int j = 1; j <= i && j <= out_len; j++) {
                const long double t = lgammal(i + 1.0L) - lgammal(j + 1.0L) - lgammal(i - j + 1.0L) + j * lp + (i - j) * lq;
                out[j - 1] += v * (double)expl(t);
            }
        } else if (faithful) {
            double l = i * prob;
            if (l == 0 || l != l)
                continue;
            long double p1 = 1;
            const long double d1 = expl(200), d2 = expl(l);
            for (int j = 1; j <= i && j <= out_len; j++) {
                p1 *= l / j;
                long double p1c = p1;
                while (l > 200 && p1c > 0) {
                    p1c /= d1;
                    l -= 200;
                }
                out[j - 1] += v * (double)(p1c / d2);
            }
        } else {
            const long double l = (long double)(i * prob), ll = logl(l);
            for (int j = 1; j <= i && j <= out_len; j++)
                out[j - 1] += v * (double)expl(j * ll - lgammal(j + 1.0L) - l);
        }
    }
}
