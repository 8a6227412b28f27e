/*
 * nc_oracle.c -- CPU restatement of nanocall's HMM hot path.  TEST INFRASTRUCTURE ONLY.
 * See nc_oracle.h for the parity status of each part.  Every function cites the reference
 * file:line it follows (paths relative to /root/reference).
 *
 * Build: gcc -std=c99 -O2 -ffp-contract=off -fPIC -shared nc_oracle.c -lm   (oracle/Makefile)
 * The reference is C++ where std::log(float) etc. select the float overloads; this file spells
 * the float/double choice out (logf vs log) wherever the reference's overload resolution does.
 */
#include "nc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------
 * Kmer  (src/nanocall/Kmer.hpp)
 * ---------------------------------------------------------------------------------------- */

/* Kmer::to_string, src/nanocall/Kmer.hpp:41-50 -- MSB pair is the first base. */
void nco_kmer_to_string(unsigned k, char out[7])
{
    static const char int_to_base[] = "ACGT";
    for (unsigned j = 0; j < NCO_KMER_SIZE; ++j)
        out[j] = int_to_base[(k >> (2 * (NCO_KMER_SIZE - j - 1))) & 0x3];
    out[NCO_KMER_SIZE] = 0;
}

/* Kmer::to_int, src/nanocall/Kmer.hpp:12-35 (unknown letters contribute -1 there; here too). */
unsigned nco_kmer_to_int(const char* s)
{
    size_t res = 0;
    for (size_t i = 0; s[i]; ++i) {
        int v;
        switch (s[i]) { case 'A': v = 0; break; case 'C': v = 1; break;
                        case 'G': v = 2; break; case 'T': v = 3; break; default: v = -1; }
        res <<= 2;
        res += (size_t)(long)v;
    }
    return (unsigned)res;
}

/* Kmer::min_skip, src/nanocall/Kmer.hpp:51-68 */
unsigned nco_kmer_min_skip(unsigned k1, unsigned k2)
{
    if (k1 == k2) return 0;
    for (unsigned k = NCO_KMER_SIZE - 1; k > 0; --k)
        if ((k1 & ((1u << (2 * k)) - 1)) == (k2 >> (2 * (NCO_KMER_SIZE - k))))
            return NCO_KMER_SIZE - k;
    return NCO_KMER_SIZE;
}

/* Kmer::prefix / suffix, src/nanocall/Kmer.hpp:69-76 */
unsigned nco_kmer_prefix(unsigned i, unsigned k) { return i >> (2 * (NCO_KMER_SIZE - k)); }
unsigned nco_kmer_suffix(unsigned i, unsigned k) { return i & ((1u << (2 * k)) - 1); }

/* Kmer::max_self_overlap, src/nanocall/Kmer.hpp:81-110 */
unsigned nco_kmer_max_self_overlap(unsigned i)
{
    for (unsigned k = NCO_KMER_SIZE - 1; k >= 1; --k)
        if (nco_kmer_suffix(i, k) == nco_kmer_prefix(i, k)) return k;
    return 0;
}

/* Kmer::neighbour_list, src/nanocall/Kmer.hpp:115-148; d==1: 4 entries, d==2: 16 entries,
 * in (b1 major, b2 minor) order == ascending. */
void nco_kmer_neighbour_list(unsigned i, unsigned d, unsigned* out)
{
    unsigned n = 0;
    for (unsigned b1 = 0; b1 < 4; ++b1) {
        unsigned i1 = (nco_kmer_suffix(i, NCO_KMER_SIZE - 1) << 2) + b1;
        if (d == 1) { out[n++] = i1; continue; }
        for (unsigned b2 = 0; b2 < 4; ++b2)
            out[n++] = (nco_kmer_suffix(i1, NCO_KMER_SIZE - 1) << 2) + b2;
    }
}

/* ------------------------------------------------------------------------------------------
 * logsumset  (hpptools include/logsumset.hpp -- ABSENT from /root/reference, no pinned SHA)
 *
 * Restated from its call sites (Forward_Backward.hpp:54,77-84,113-120,129-134;
 * Parameter_Trainer.hpp:441-517; State_Transitions.hpp:87-102): a multiset of log-values whose
 * val() folds the two smallest, a <= b, into b + log1p(exp(a - b)) until one value is left;
 * empty set -> -INF (forced by State_Transitions.hpp:93).  PARITY UNPINNED; every use is
 * tolerance-checked (1e-4 relative), never bit-checked.
 * ---------------------------------------------------------------------------------------- */
static void heap_sift_down(float* h, size_t n, size_t i)
{
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && h[l] < h[m]) m = l;
        if (r < n && h[r] < h[m]) m = r;
        if (m == i) return;
        float t = h[i]; h[i] = h[m]; h[m] = t;
        i = m;
    }
}

float nco_logsumset_val(float* h, size_t n)
{
    if (n == 0) return -INFINITY;
    for (size_t i = n / 2; i-- > 0;) heap_sift_down(h, n, i);
    while (n > 1) {
        float a = h[0];                  /* smallest */
        h[0] = h[--n]; heap_sift_down(h, n, 0);
        float b = h[0];                  /* second smallest, a <= b */
        float r = (a == -INFINITY) ? b : b + log1pf(expf(a - b));
        h[0] = r; heap_sift_down(h, n, 0);
    }
    return h[0];
}

typedef struct { float* v; size_t n, cap; } lss;
static void lss_add(lss* s, float x)
{
    if (s->n == s->cap) {
        s->cap = s->cap ? 2 * s->cap : 64;
        s->v = (float*)realloc(s->v, s->cap * sizeof(float));
    }
    s->v[s->n++] = x;
}
static float lss_val(lss* s) { float r = nco_logsumset_val(s->v, s->n); s->n = 0; return r; }
static void lss_free(lss* s) { free(s->v); s->v = NULL; s->n = s->cap = 0; }

/* ------------------------------------------------------------------------------------------
 * State_Transitions  (src/nanocall/State_Transitions.hpp)
 * ---------------------------------------------------------------------------------------- */

/* State_Transitions::get_trans_prob, src/nanocall/State_Transitions.hpp:125-144.
 * `p` is float; pow() is the double function, so each skip term is added in double and the sum
 * rounded back to float (compound assignment), exactly as the C++ does. */
float nco_trans_prob(unsigned i, unsigned j, float p_stay, float p_step, float p_skip_1)
{
    float p = 0;
    if (i == j) p += p_stay;
    if (nco_kmer_suffix(i, NCO_KMER_SIZE - 1) == nco_kmer_prefix(j, NCO_KMER_SIZE - 1))
        p += p_step / 4;
    for (unsigned l = 2; l < NCO_KMER_SIZE; ++l)
        if (nco_kmer_suffix(i, NCO_KMER_SIZE - l) == nco_kmer_prefix(j, NCO_KMER_SIZE - l))
            p = (float)((double)p + pow((double)p_skip_1, (double)(l - 1)) / (double)(1u << (2 * l)));
    p = (float)((double)p
                + (pow((double)p_skip_1, 5.0) / (double)(1.0f - p_skip_1)) / (double)NCO_N_STATES);
    return p;
}

static int cmp_u32(const void* a, const void* b)
{
    uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return (x > y) - (x < y);
}

/* State_Transitions::compute_transitions_fast, src/nanocall/State_Transitions.hpp:181-224, and
 * update_fields, :79-104 (from_v = inversion of to_v scanning i ascending).  p_rest_from/to
 * (:87-102) are never read on the hot path and are not restated. */
nco_transitions* nco_transitions_fast(float p_skip, float p_stay)
{
    nco_transitions* t = (nco_transitions*)calloc(1, sizeof(*t));
    uint32_t* to_idx = (uint32_t*)malloc(sizeof(uint32_t) * NCO_N_STATES * 21);
    float* to_w = (float*)malloc(sizeof(float) * NCO_N_STATES * 21);
    /* :198-202 -- both in double, rounded to float on assignment */
    float p_step = (float)(1.0 - (double)p_stay - (double)p_skip);
    float p_skip_1 = (float)((double)p_skip / ((double)p_skip + 1.0));
    uint32_t n = 0;
    for (unsigned i = 0; i < NCO_N_STATES; ++i) {
        /* std::set<unsigned> to_s{i} + nl1 + nl2 (:208-212): sorted, de-duplicated */
        uint32_t s[21]; unsigned m = 0, nl[16];
        s[m++] = i;
        nco_kmer_neighbour_list(i, 1, nl); for (unsigned k = 0; k < 4; ++k) s[m++] = nl[k];
        nco_kmer_neighbour_list(i, 2, nl); for (unsigned k = 0; k < 16; ++k) s[m++] = nl[k];
        qsort(s, m, sizeof(uint32_t), cmp_u32);
        t->to_ptr[i] = n;
        for (unsigned k = 0; k < m; ++k) {
            if (k > 0 && s[k] == s[k - 1]) continue;
            float p = nco_trans_prob(i, s[k], p_stay, p_step, p_skip_1);
            to_idx[n] = s[k];
            to_w[n] = logf(p); /* std::log(Float_Type) :216 */
            ++n;
        }
    }
    t->to_ptr[NCO_N_STATES] = n;
    t->n_arcs = n;
    t->to_idx = to_idx; t->to_logw = to_w;
    /* update_fields :85-94 */
    t->from_idx = (uint32_t*)malloc(sizeof(uint32_t) * n);
    t->from_logw = (float*)malloc(sizeof(float) * n);
    uint32_t* cnt = (uint32_t*)calloc(NCO_N_STATES + 1, sizeof(uint32_t));
    for (uint32_t a = 0; a < n; ++a) cnt[to_idx[a] + 1]++;
    for (unsigned j = 0; j < NCO_N_STATES; ++j) cnt[j + 1] += cnt[j];
    memcpy(t->from_ptr, cnt, sizeof(uint32_t) * (NCO_N_STATES + 1));
    for (unsigned i = 0; i < NCO_N_STATES; ++i)
        for (uint32_t a = t->to_ptr[i]; a < t->to_ptr[i + 1]; ++a) {
            uint32_t pos = cnt[to_idx[a]]++;
            t->from_idx[pos] = i;
            t->from_logw[pos] = to_w[a];
        }
    free(cnt);
    return t;
}

void nco_transitions_free(nco_transitions* t)
{
    if (!t) return;
    free(t->from_idx); free(t->from_logw); free(t->to_idx); free(t->to_logw); free(t);
}

uint32_t nco_transitions_n_arcs(const nco_transitions* t) { return t->n_arcs; }

void nco_transitions_export_from(const nco_transitions* t, uint32_t* row_ptr, uint32_t* idx, float* logw)
{
    memcpy(row_ptr, t->from_ptr, sizeof(t->from_ptr));
    memcpy(idx, t->from_idx, sizeof(uint32_t) * t->n_arcs);
    memcpy(logw, t->from_logw, sizeof(float) * t->n_arcs);
}

void nco_transitions_export_to(const nco_transitions* t, uint32_t* row_ptr, uint32_t* idx, float* logw)
{
    memcpy(row_ptr, t->to_ptr, sizeof(t->to_ptr));
    memcpy(idx, t->to_idx, sizeof(uint32_t) * t->n_arcs);
    memcpy(logw, t->to_logw, sizeof(float) * t->n_arcs);
}

/* ------------------------------------------------------------------------------------------
 * Pore_Model  (src/nanocall/Pore_Model.hpp)
 * ---------------------------------------------------------------------------------------- */

/* alg::mean_stdv_of is in hpptools alg.hpp (ABSENT). Used only for Pore_Model::mean()/stdv()
 * (Pore_Model.hpp:307-313), off the DP path.  Restated as mean and sample (n-1) stdv in float;
 * PARITY UNPINNED. */
static void model_update_statistics(nco_model* m)
{
    float s = 0, s2 = 0;
    for (unsigned i = 0; i < NCO_N_STATES; ++i) {
        s += m->st[i].level_mean;
        s2 += m->st[i].level_mean * m->st[i].level_mean;
    }
    float n = (float)NCO_N_STATES;
    m->mean = s / n;
    float var = (s2 - s * m->mean) / (n - 1);
    m->stdv = var > 0 ? sqrtf(var) : 0;
}

/* Pore_Model::load_from_vector, src/nanocall/Pore_Model.hpp:220-239;
 * update_sd_lambda :112 (double pow, rounded to float); update_logs :118-124 (float log). */
void nco_model_load_from_vector(nco_model* m, const float* v)
{
    for (unsigned i = 0; i < NCO_N_STATES; ++i) {
        nco_state* s = &m->st[i];
        s->level_mean = v[4 * i + 0];
        s->level_stdv = v[4 * i + 1];
        s->sd_mean = v[4 * i + 2];
        s->sd_stdv = v[4 * i + 3];
        s->sd_lambda = (float)(pow((double)s->sd_mean, 3.0) / pow((double)s->sd_stdv, 2.0));
        s->log_level_mean = logf(s->level_mean);
        s->log_level_stdv = logf(s->level_stdv);
        s->log_sd_mean = logf(s->sd_mean);
        s->log_sd_stdv = 0; /* never set by the reference (:118-124), never read */
        s->log_sd_lambda = logf(s->sd_lambda);
    }
    model_update_statistics(m);
}

/* Pore_Model::scale, src/nanocall/Pore_Model.hpp:190-201, and Pore_Model_State::scale :126-138:
 * logs of stdv/sd_mean/sd_lambda are ADDED to, only log_level_mean is recomputed. */
void nco_model_scale(nco_model* m, const nco_pm_params* p)
{
    float log_var = logf(p->var), log_scale_sd = logf(p->scale_sd), log_var_sd = logf(p->var_sd);
    for (unsigned i = 0; i < NCO_N_STATES; ++i) {
        nco_state* s = &m->st[i];
        s->level_mean = s->level_mean * p->scale + p->shift;
        s->level_stdv = s->level_stdv * p->var;
        s->sd_mean = s->sd_mean * p->scale_sd;
        s->sd_lambda = s->sd_lambda * p->var_sd;
        /* update_sd_stdv :115 */
        s->sd_stdv = (float)pow(pow((double)s->sd_mean, 3.0) / (double)s->sd_lambda, .5);
        s->log_level_mean = logf(s->level_mean);
        s->log_level_stdv += log_var;
        s->log_sd_mean += log_scale_sd;
        s->log_sd_lambda += log_var_sd;
    }
    model_update_statistics(m);
}

void nco_model_export6(const nco_model* m, float* out)
{
    for (unsigned i = 0; i < NCO_N_STATES; ++i) {
        const nco_state* s = &m->st[i];
        out[6 * i + 0] = s->level_mean;
        out[6 * i + 1] = s->level_stdv;
        out[6 * i + 2] = s->log_level_stdv;
        out[6 * i + 3] = s->sd_mean;
        out[6 * i + 4] = s->sd_lambda;
        out[6 * i + 5] = s->log_sd_lambda;
    }
}

/* log_normal_pdf, src/nanocall/Pore_Model.hpp:24-31 */
static float log_normal_pdf(float x, float mean, float stdv, float log_stdv)
{
    const float log_2pi = (float)log(2.0 * M_PI);
    float a = (x - mean) / stdv;
    return -log_stdv - (log_2pi + a * a) / 2.0f;
}

/* log_invgauss_pdf, src/nanocall/Pore_Model.hpp:33-40 */
static float log_invgauss_pdf(float x, float log_x, float mu, float lambda, float log_lambda)
{
    const float log_2pi = (float)log(2.0 * M_PI);
    float a = (x - mu) / mu;
    return (log_lambda - log_2pi - 3.0f * log_x - lambda * a * a / x) / 2.0f;
}

/* Pore_Model_State::log_pr_corrected_emission, src/nanocall/Pore_Model.hpp:145-149 */
float nco_log_pr_corrected_emission(const nco_state* s, float corrected_mean, float stdv, float log_stdv)
{
    return log_normal_pdf(corrected_mean, s->level_mean, s->level_stdv, s->log_level_stdv)
         + log_invgauss_pdf(stdv, log_stdv, s->sd_mean, s->sd_lambda, s->log_sd_lambda);
}

/* ------------------------------------------------------------------------------------------
 * Event  (src/nanocall/Event.hpp)
 * ---------------------------------------------------------------------------------------- */

/* Event operator>> :59-68 + update_logs :35-45 (stdv == 0 -> 0.01) */
void nco_event_init(nco_event* e, float mean, float stdv, float start, float length)
{
    e->mean = mean; e->stdv = stdv; e->start = start; e->length = length;
    e->corrected_mean = mean;
    e->log_mean = logf(e->mean);
    e->log_corrected_mean = logf(e->corrected_mean);
    if (e->stdv == 0.0) e->stdv = (float)0.01;
    e->log_stdv = logf(e->stdv);
    e->model_state_idx = 0; e->move = 0;
}

/* Event_Sequence::apply_drift_correction, src/nanocall/Event.hpp:77-84 */
void nco_events_apply_drift_correction(nco_event* ev, size_t n, float drift)
{
    for (size_t i = 0; i < n; ++i) {
        ev[i].corrected_mean -= drift * ev[i].start;
        ev[i].log_corrected_mean = logf(ev[i].corrected_mean);
    }
}

/* Event_Sequence::get_base_seq, src/nanocall/Event.hpp:85-99 */
size_t nco_events_get_base_seq(const nco_event* ev, size_t n, char* out)
{
    char km[7];
    size_t len = 0;
    if (n == 0) { out[0] = 0; return 0; }
    nco_kmer_to_string(ev[0].model_state_idx, km);
    memcpy(out, km, NCO_KMER_SIZE); len = NCO_KMER_SIZE;
    for (size_t i = 1; i < n; ++i) {
        unsigned a = (unsigned)ev[i].move < NCO_KMER_SIZE ? (unsigned)ev[i].move : NCO_KMER_SIZE;
        unsigned b = NCO_KMER_SIZE - a;
        nco_kmer_to_string(ev[i].model_state_idx, km);
        memcpy(out + len, km + b, a); len += a;
    }
    out[len] = 0;
    return len;
}

/* write_fasta, src/nanocall/nanocall.cpp:584-591 */
size_t nco_write_fasta(char* out, size_t cap, const char* name, const char* seq, unsigned line_width)
{
    size_t n = 0, L = strlen(seq), nl = strlen(name);
    if (cap < nl + L + L / (line_width ? line_width : 1) + 8) return 0;
    out[n++] = '>'; memcpy(out + n, name, nl); n += nl; out[n++] = '\n';
    for (size_t pos = 0; pos < L; pos += line_width) {
        size_t c = L - pos < line_width ? L - pos : line_width;
        memcpy(out + n, seq + pos, c); n += c; out[n++] = '\n';
    }
    out[n] = 0;
    return n;
}

/* ------------------------------------------------------------------------------------------
 * Viterbi  (src/nanocall/Viterbi.hpp)
 * ---------------------------------------------------------------------------------------- */
typedef struct { float alpha; unsigned beta; } vit_cell; /* Viterbi.hpp:26-30 */

/* Viterbi::fill :44-99, fill_state_seq :120-142, fill_move_seq :144-150 */
float nco_viterbi_fill(const nco_model* pm, const nco_transitions* st, nco_event* ev, size_t n)
{
    if (n == 0) return NAN;
    vit_cell* m = (vit_cell*)malloc(sizeof(vit_cell) * NCO_N_STATES * n); /* :50 */
    if (!m) return NAN;
    float log_n_states = logf((float)NCO_N_STATES); /* :51 */
#define CELL(i, j) m[(size_t)(i) * NCO_N_STATES + (j)]
    for (unsigned j = 0; j < NCO_N_STATES; ++j) { /* :55-68 */
        CELL(0, j).alpha = nco_log_pr_corrected_emission(&pm->st[j], ev[0].corrected_mean, ev[0].stdv,
                                                         ev[0].log_stdv) - log_n_states;
        CELL(0, j).beta = NCO_N_STATES;
    }
    for (size_t i = 1; i < n; ++i) { /* :72-96 */
        for (unsigned j = 0; j < NCO_N_STATES; ++j) {
            float a = -INFINITY; unsigned b = NCO_N_STATES;
            for (uint32_t k = st->from_ptr[j]; k < st->from_ptr[j + 1]; ++k) {
                unsigned j_prev = st->from_idx[k];
                float v = st->from_logw[k] + CELL(i - 1, j_prev).alpha;
                if (v > a) { a = v; b = j_prev; } /* strict >, ascending j_prev: lowest index wins ties */
            }
            a += nco_log_pr_corrected_emission(&pm->st[j], ev[i].corrected_mean, ev[i].stdv, ev[i].log_stdv);
            CELL(i, j).alpha = a; CELL(i, j).beta = b;
        }
    }
    /* fill_state_seq :120-142 */
    float max_v = -INFINITY; unsigned max_j = NCO_N_STATES;
    for (unsigned j = 0; j < NCO_N_STATES; ++j)
        if (CELL(n - 1, j).alpha > max_v) { max_j = j; max_v = CELL(n - 1, j).alpha; }
    float path_probability = max_v;
    if (max_j == NCO_N_STATES) { free(m); return NAN; } /* reference reads out of row here; guard */
    for (size_t i = n - 1; i > 0; --i) {
        ev[i].model_state_idx = max_j;
        max_j = CELL(i, max_j).beta;
        if (max_j == NCO_N_STATES) { free(m); return NAN; }
    }
    ev[0].model_state_idx = max_j;
    /* fill_move_seq :144-150 */
    for (size_t i = 0; i < n; ++i)
        ev[i].move = i > 0 ? (int)nco_kmer_min_skip(ev[i - 1].model_state_idx, ev[i].model_state_idx) : 0;
#undef CELL
    free(m);
    return path_probability;
}

/* How many cells (i >= 1, j) of Viterbi::fill's matrix have their maximum attained by TWO OR MORE predecessors with exactly
 * equal floats (Viterbi.hpp:79-89: the ascending strict-> scan then keeps the lowest index).  SURVEY.md section 0.7 records this
 * count from a probe of the real reference (866 of 12.3 M cells on a 3 000-event read); tests/test_reference_pins.py compares.
 * Two rows of alpha only: the count needs no back-pointers. */
uint64_t nco_viterbi_tie_cells(const nco_model* pm, const nco_transitions* st, size_t n, const float* corrected_mean,
                               const float* stdv, const float* log_stdv)
{
    if (n == 0) return 0;
    float* prev = (float*)malloc(sizeof(float) * NCO_N_STATES);
    float* cur = (float*)malloc(sizeof(float) * NCO_N_STATES);
    if (!prev || !cur) { free(prev); free(cur); return (uint64_t)-1; }
    const float log_n_states = logf((float)NCO_N_STATES);
    for (unsigned j = 0; j < NCO_N_STATES; ++j)
        prev[j] = nco_log_pr_corrected_emission(&pm->st[j], corrected_mean[0], stdv[0], log_stdv[0]) - log_n_states;
    uint64_t ties = 0;
    for (size_t i = 1; i < n; ++i) {
        for (unsigned j = 0; j < NCO_N_STATES; ++j) {
            float a = -INFINITY; unsigned at_max = 0;
            for (uint32_t k = st->from_ptr[j]; k < st->from_ptr[j + 1]; ++k) {
                const float v = st->from_logw[k] + prev[st->from_idx[k]];
                if (v > a) { a = v; at_max = 1; } else if (v == a) ++at_max;
            }
            ties += at_max > 1;
            cur[j] = a + nco_log_pr_corrected_emission(&pm->st[j], corrected_mean[i], stdv[i], log_stdv[i]);
        }
        float* t = prev; prev = cur; cur = t;
    }
    free(prev); free(cur);
    return ties;
}

float nco_viterbi_soa(const nco_model* pm, const nco_transitions* st, size_t n,
                      const float* corrected_mean, const float* stdv, const float* log_stdv,
                      uint16_t* out_state, int32_t* out_move)
{
    nco_event* ev = (nco_event*)calloc(n ? n : 1, sizeof(nco_event));
    for (size_t i = 0; i < n; ++i) {
        ev[i].corrected_mean = corrected_mean[i]; ev[i].stdv = stdv[i]; ev[i].log_stdv = log_stdv[i];
    }
    float r = nco_viterbi_fill(pm, st, ev, n);
    for (size_t i = 0; i < n; ++i) {
        if (out_state) out_state[i] = (uint16_t)ev[i].model_state_idx;
        if (out_move) out_move[i] = ev[i].move;
    }
    free(ev);
    return r;
}

/* ------------------------------------------------------------------------------------------
 * Forward_Backward  (src/nanocall/Forward_Backward.hpp)
 * ---------------------------------------------------------------------------------------- */

/* Forward_Backward::fill :46-135 */
nco_fwbw* nco_fwbw_fill(const nco_model* pm, const nco_transitions* st, const nco_event* ev, size_t n)
{
    if (n == 0) return NULL;
    nco_fwbw* f = (nco_fwbw*)calloc(1, sizeof(*f));
    f->n_events = n;
    f->alpha = (float*)calloc(n * NCO_N_STATES, sizeof(float)); /* resize() zero-fills :50-52 */
    f->beta = (float*)calloc(n * NCO_N_STATES, sizeof(float));
    float log_n_states = logf((float)NCO_N_STATES);
    lss s = {0};
#define A(i, j) f->alpha[(size_t)(i) * NCO_N_STATES + (j)]
#define B(i, j) f->beta[(size_t)(i) * NCO_N_STATES + (j)]
#define EMIS(j, e) nco_log_pr_corrected_emission(&pm->st[j], (e).corrected_mean, (e).stdv, (e).log_stdv)
    for (unsigned j = 0; j < NCO_N_STATES; ++j) /* :58-68 */
        A(0, j) = EMIS(j, ev[0]) - log_n_states;
    for (size_t i = 1; i < n; ++i) /* :72-89 */
        for (unsigned j = 0; j < NCO_N_STATES; ++j) {
            for (uint32_t k = st->from_ptr[j]; k < st->from_ptr[j + 1]; ++k)
                lss_add(&s, st->from_logw[k] + A(i - 1, st->from_idx[k]));
            float v = lss_val(&s);
            A(i, j) = EMIS(j, ev[i]) + v;
        }
    for (unsigned j = 0; j < NCO_N_STATES; ++j) B(n - 1, j) = 0; /* :93-103 */
    for (size_t ip1 = n - 1; ip1 > 0; --ip1) { /* :107-125 */
        size_t i = ip1 - 1;
        for (unsigned j = 0; j < NCO_N_STATES; ++j) {
            for (uint32_t k = st->to_ptr[j]; k < st->to_ptr[j + 1]; ++k) {
                unsigned j_next = st->to_idx[k];
                lss_add(&s, st->to_logw[k] + EMIS(j_next, ev[ip1]) + B(ip1, j_next)); /* :118 */
            }
            B(i, j) += lss_val(&s); /* `+=` onto a zero cell :120 */
        }
    }
    for (unsigned j = 0; j < NCO_N_STATES; ++j) lss_add(&s, A(n - 1, j)); /* :129-134 */
    f->log_pr_data = lss_val(&s);
    lss_free(&s);
    return f;
}

void nco_fwbw_free(nco_fwbw* f)
{
    if (!f) return;
    free(f->alpha); free(f->beta); free(f);
}

float nco_fwbw_soa(const nco_model* pm, const nco_transitions* st, size_t n,
                   const float* corrected_mean, const float* stdv, const float* log_stdv,
                   float* out_alpha, float* out_beta)
{
    nco_event* ev = (nco_event*)calloc(n ? n : 1, sizeof(nco_event));
    for (size_t i = 0; i < n; ++i) {
        ev[i].corrected_mean = corrected_mean[i]; ev[i].stdv = stdv[i]; ev[i].log_stdv = log_stdv[i];
    }
    nco_fwbw* f = nco_fwbw_fill(pm, st, ev, n);
    free(ev);
    if (!f) return NAN;
    if (out_alpha) memcpy(out_alpha, f->alpha, sizeof(float) * n * NCO_N_STATES);
    if (out_beta) memcpy(out_beta, f->beta, sizeof(float) * n * NCO_N_STATES);
    float r = f->log_pr_data;
    nco_fwbw_free(f);
    return r;
}

/* ------------------------------------------------------------------------------------------
 * Parameter_Trainer  (src/nanocall/Parameter_Trainer.hpp)
 * ---------------------------------------------------------------------------------------- */

/* Parameter_Trainer::init :30-57 */
unsigned nco_st_train_kmers(unsigned* out)
{
    unsigned n = 0;
    for (unsigned i = 0; i < NCO_N_STATES; ++i) {
        if (nco_kmer_max_self_overlap(i) > 0) continue;
        int all_good = 1;
        for (unsigned b1 = 0; b1 < 4; ++b1) {
            unsigned j = (nco_kmer_suffix(i, NCO_KMER_SIZE - 1) << 2) + b1;
            if (nco_kmer_max_self_overlap(j) > 1) { all_good = 0; break; }
        }
        if (all_good) out[n++] = i;
    }
    return n;
}

typedef struct {
    const nco_train_input* in;
    const nco_pm_params* pm_params;
    const nco_st_params* st_params[2];
    nco_model* scaled_model[2];
    nco_transitions* transitions[2];
    nco_event** corrected;
    nco_fwbw** fwbw;
    float fit;
} train_data;

static float log_posterior(const nco_fwbw* f, size_t i, unsigned j)
{   /* Forward_Backward::log_posterior, Forward_Backward.hpp:41 */
    return f->alpha[i * NCO_N_STATES + j] + f->beta[i * NCO_N_STATES + j] - f->log_pr_data;
}

/* fill_train_data :99-155 */
static void fill_train_data(train_data* d)
{
    const nco_train_input* in = d->in;
    int init_m[2] = {0, 0};
    for (size_t k = 0; k < in->n_seqs; ++k) { /* :105-114 */
        unsigned st = in->seq_strand[k];
        if (init_m[st]) continue;
        d->scaled_model[st] = (nco_model*)malloc(sizeof(nco_model));
        memcpy(d->scaled_model[st], in->model[st], sizeof(nco_model));
        nco_model_scale(d->scaled_model[st], d->pm_params);
        init_m[st] = 1;
    }
    int init_t[2] = {0, 0};
    for (size_t k = 0; k < in->n_seqs; ++k) { /* :119-133; is_default() State_Transitions.hpp:34-37 */
        unsigned st = in->seq_strand[k];
        if (init_t[st]) continue;
        const nco_st_params* sp = d->st_params[st];
        if (!(sp->p_stay == in->default_p_stay && sp->p_skip == in->default_p_skip))
            d->transitions[st] = nco_transitions_fast(sp->p_skip, sp->p_stay);
        else
            d->transitions[st] = nco_transitions_fast(in->default_p_skip, in->default_p_stay);
        init_t[st] = 1;
    }
    d->corrected = (nco_event**)calloc(in->n_seqs, sizeof(nco_event*));
    d->fwbw = (nco_fwbw**)calloc(in->n_seqs, sizeof(nco_fwbw*));
    d->fit = 0.0f;
    for (size_t k = 0; k < in->n_seqs; ++k) { /* :141-155 */
        unsigned st = in->seq_strand[k];
        size_t n = in->seq_len[k];
        d->corrected[k] = (nco_event*)malloc(sizeof(nco_event) * (n ? n : 1));
        memcpy(d->corrected[k], in->seqs[k], sizeof(nco_event) * n);
        nco_events_apply_drift_correction(d->corrected[k], n, d->pm_params->drift);
        d->fwbw[k] = nco_fwbw_fill(d->scaled_model[st], d->transitions[st], d->corrected[k], n);
        d->fit += d->fwbw[k]->log_pr_data;
    }
}

static void free_train_data(train_data* d)
{
    for (size_t k = 0; k < d->in->n_seqs; ++k) { free(d->corrected[k]); nco_fwbw_free(d->fwbw[k]); }
    free(d->corrected); free(d->fwbw);
    for (int st = 0; st < 2; ++st) { free(d->scaled_model[st]); nco_transitions_free(d->transitions[st]); }
}

/* train_pm_params :230-427 */
static void train_pm_params(const train_data* d, nco_pm_params* np, int* done)
{
    const nco_train_input* in = d->in;
    *done = 0;
    unsigned total_n_events = 0;
    double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    double B[3] = {0, 0, 0};
    double D = 0, V_numer = 0, V_denom = 0, U_pos = 0;
    for (size_t k = 0; k < in->n_seqs; ++k) {
        unsigned st = in->seq_strand[k];
        const nco_event* events = in->seqs[k];      /* UNcorrected events :263 */
        size_t n = in->seq_len[k];
        total_n_events += (unsigned)n;
        const nco_model* pm = in->model[st];        /* UNscaled model :266 */
        const nco_fwbw* f = d->fwbw[k];
        for (size_t i = 0; i < n; ++i) {
            float x_i = events[i].mean, y_i = events[i].stdv, t_i = events[i].start;
            float s[3] = {0, 0, 0}, l[3] = {0, 0, 0};
            for (unsigned j = 0; j < NCO_N_STATES; ++j) { /* :282-303 */
                const nco_state* q = &pm->st[j];
                float p_ij = expf(log_posterior(f, i, j));
                float term_s0 = p_ij / (q->level_stdv * q->level_stdv);
                float term_s1 = term_s0 * q->level_mean;
                float term_s2 = term_s1 * q->level_mean;
                float term_l0 = p_ij * q->sd_lambda;
                float term_l1 = term_l0 / q->sd_mean;
                float term_l2 = term_l1 / q->sd_mean;
                s[0] += term_s0; s[1] += term_s1; s[2] += term_s2;
                l[0] += term_l0; l[1] += term_l1; l[2] += term_l2;
            }
            /* :304-320 -- float products, accumulated in double */
            A[0][0] += s[0];
            A[0][1] += s[1];
            A[1][1] += s[2];
            B[0] += s[0] * x_i;
            B[1] += s[1] * x_i;
            if (in->train_drift) {
                A[0][2] += s[0] * t_i;
                A[1][2] += s[1] * t_i;
                A[2][2] += s[0] * t_i * t_i;
                B[2] += s[0] * x_i * t_i;
            }
            D += s[0] * x_i * x_i;
            V_numer += l[2] * y_i;
            V_denom += l[1];
            U_pos += l[0] / y_i;
        }
    }
    A[1][0] = A[0][1]; A[2][0] = A[0][2]; A[2][1] = A[1][2];
    if (!in->train_drift) A[2][2] = 1.0;
    double Ac[3][3], Bc[3], C[3];
    memcpy(Ac, A, sizeof(A)); memcpy(Bc, B, sizeof(B));
    for (unsigned i = 0; i < 3; ++i) { /* alg::max_value_of :328 */
        C[i] = A[i][0];
        if (A[i][1] > C[i]) C[i] = A[i][1];
        if (A[i][2] > C[i]) C[i] = A[i][2];
    }
    for (unsigned i = 0; i < 3; ++i) { /* :340-386 */
        unsigned p = i;
        double p_val = fabs(A[i][i]) / C[p];
        for (unsigned i2 = i + 1; i2 < 3; ++i2) {
            double i2_val = fabs(A[i2][i]) / C[i2];
            if (i2_val > p_val) { p = i2; p_val = i2_val; }
        }
        if (p_val < 1e-7) { *done = 1; *np = *d->pm_params; return; } /* :355-360 */
        if (p > i) {
            for (unsigned c = 0; c < 3; ++c) { double t = A[i][c]; A[i][c] = A[p][c]; A[p][c] = t; }
            double t = B[i]; B[i] = B[p]; B[p] = t;
            t = C[i]; C[i] = C[p]; C[p] = t;
        }
        for (p = i + 1; p < 3; ++p) {
            double mm = A[p][i] / A[i][i];
            A[p][i] = 0;
            for (unsigned j = i + 1; j < 3; ++j) A[p][j] -= mm * A[i][j];
            B[p] -= mm * B[i];
        }
    }
    /* :388-390 -- each solution is stored into a float member before the next one reads it */
    np->drift = (float)(B[2] / A[2][2]);
    np->scale = (float)((B[1] - A[1][2] * np->drift) / A[1][1]);
    np->shift = (float)((B[0] - A[0][1] * np->scale - A[0][2] * np->drift) / A[0][0]);
    float a_hat = np->shift, b_hat = np->scale, c_hat = np->drift;
    /* :406-416 -- note a_hat*a_hat etc. are float products, the 2.0* terms are double */
    double d_numer = (D
                      + (double)(a_hat * a_hat) * Ac[0][0]
                      + (double)(b_hat * b_hat) * Ac[1][1]
                      + (double)(c_hat * c_hat) * Ac[2][2]
                      + 2.0 * a_hat * b_hat * Ac[0][1]
                      + 2.0 * a_hat * c_hat * Ac[0][2]
                      + 2.0 * b_hat * c_hat * Ac[1][2]
                      - 2.0 * (a_hat * Bc[0] + b_hat * Bc[1] + c_hat * Bc[2]));
    np->var = (float)sqrt(d_numer / (double)total_n_events);             /* :417 */
    np->scale_sd = (float)(V_numer / V_denom);                            /* :422 */
    np->var_sd = (float)((double)total_n_events / (U_pos - V_denom / np->scale_sd)); /* :426 */
}

/* train_st_params :434-532 */
static void train_st_params(const train_data* d, nco_st_params new_st[2])
{
    const nco_train_input* in = d->in;
    unsigned kmers[NCO_N_STATES];
    unsigned n_kmers = nco_st_train_kmers(kmers);
    for (unsigned st = 0; st < 2; ++st) {
        lss s_stay = {0}, s_skip = {0}, s_denom = {0}, s2 = {0};
        float log_p_stay = logf(d->st_params[st]->p_stay);
        float log_p_step_4 = (float)(log(1.0 - (double)d->st_params[st]->p_stay - (double)d->st_params[st]->p_skip)
                                     - log(4.0)); /* :444-445 */
        for (size_t k = 0; k < in->n_seqs; ++k) {
            if (in->seq_strand[k] != st) continue;
            const nco_model* spm = d->scaled_model[st];
            const nco_event* ce = d->corrected[k];
            size_t n = in->seq_len[k];
            const nco_fwbw* f = d->fwbw[k];
#define JOINT(i, j1, j2, lpt) \
    (f->alpha[(i) * NCO_N_STATES + (j1)] + (lpt) \
     + nco_log_pr_corrected_emission(&spm->st[j2], ce[(i) + 1].corrected_mean, ce[(i) + 1].stdv, ce[(i) + 1].log_stdv) \
     + f->beta[((i) + 1) * NCO_N_STATES + (j2)] - f->log_pr_data) /* :456-462 */
            for (size_t i = 0; i + 1 < n; ++i) {
                for (unsigned a = 0; a < n_kmers; ++a) {
                    unsigned j1 = kmers[a];
                    float log_p_j1 = log_posterior(f, i, j1);
                    lss_add(&s_denom, log_p_j1);
                    float log_p_j1_j1 = JOINT(i, j1, j1, log_p_stay);
                    if (log_p_j1_j1 > log_p_j1) log_p_j1_j1 = log_p_j1; /* :480-488 */
                    lss_add(&s_stay, log_p_j1_j1);
                    float log_p_j1_d01;
                    {
                        unsigned nl[4];
                        lss_add(&s2, log_p_j1_j1);
                        nco_kmer_neighbour_list(j1, 1, nl);
                        for (unsigned b = 0; b < 4; ++b) lss_add(&s2, JOINT(i, j1, nl[b], log_p_step_4));
                        log_p_j1_d01 = lss_val(&s2);
                    }
                    if (log_p_j1_d01 > log_p_j1) log_p_j1_d01 = log_p_j1; /* :502-510 */
                    float p_j1_d2 = expf(log_p_j1) - expf(log_p_j1_d01);
                    lss_add(&s_skip, logf(p_j1_d2)); /* :511-512 */
                }
            }
#undef JOINT
        }
        float denom = lss_val(&s_denom);
        new_st[st].p_stay = expf(lss_val(&s_stay) - denom); /* :516-517 */
        new_st[st].p_skip = expf(lss_val(&s_skip) - denom);
        /* :518-530 -- comparisons against the DOUBLE literals .05 / .4 */
        if ((double)new_st[st].p_stay < .05 || (double)new_st[st].p_stay > .4
            || (double)new_st[st].p_skip < .05 || (double)new_st[st].p_skip > .4) {
            nco_st_params alt; /* std::max(a, b) = a < b ? b : a ; std::min(a, b) = b < a ? b : a */
            alt.p_stay = new_st[st].p_stay < .05f ? .05f : new_st[st].p_stay;
            alt.p_stay = .4f < alt.p_stay ? .4f : alt.p_stay;
            alt.p_skip = new_st[st].p_skip < .05f ? .05f : new_st[st].p_skip;
            alt.p_skip = .4f < alt.p_skip ? .4f : alt.p_skip;
            new_st[st] = alt;
        }
        lss_free(&s_stay); lss_free(&s_skip); lss_free(&s_denom); lss_free(&s2);
    }
}

/* train_one_round :541-579 */
float nco_train_one_round(const nco_train_input* in,
                          const nco_pm_params* crt_pm, const nco_st_params crt_st[2],
                          nco_pm_params* new_pm, nco_st_params new_st[2],
                          int* done, int train_scaling, int train_transitions)
{
    train_data d;
    memset(&d, 0, sizeof(d));
    d.in = in; d.pm_params = crt_pm; d.st_params[0] = &crt_st[0]; d.st_params[1] = &crt_st[1];
    *done = 0;
    fill_train_data(&d);
    float fit = d.fit;
    if (train_scaling) {
        train_pm_params(&d, new_pm, done);
        if (*done) { new_st[0] = crt_st[0]; new_st[1] = crt_st[1]; free_train_data(&d); return fit; }
    }
    if (train_transitions) train_st_params(&d, new_st);
    free_train_data(&d);
    return fit;
}

float nco_train_one_round_soa(size_t n_seqs, const uint64_t* off, const unsigned* strand,
                              const float* mean, const float* stdv, const float* start,
                              const float* model0, const float* model1,
                              float default_p_stay, float default_p_skip, int train_drift,
                              const float crt_pm[6], const float crt_st[4],
                              float new_pm[6], float new_st[4], int* done,
                              int train_scaling, int train_transitions)
{
    nco_train_input in;
    memset(&in, 0, sizeof(in));
    nco_event** seqs = (nco_event**)calloc(n_seqs, sizeof(nco_event*));
    size_t* len = (size_t*)calloc(n_seqs, sizeof(size_t));
    for (size_t k = 0; k < n_seqs; ++k) {
        len[k] = (size_t)(off[k + 1] - off[k]);
        seqs[k] = (nco_event*)calloc(len[k] ? len[k] : 1, sizeof(nco_event));
        for (size_t i = 0; i < len[k]; ++i)
            nco_event_init(&seqs[k][i], mean[off[k] + i], stdv[off[k] + i], start[off[k] + i], 0.0f);
    }
    nco_model* m0 = (nco_model*)malloc(sizeof(nco_model));
    nco_model* m1 = (nco_model*)malloc(sizeof(nco_model));
    nco_model_load_from_vector(m0, model0);
    nco_model_load_from_vector(m1, model1 ? model1 : model0);
    in.n_seqs = n_seqs; in.seqs = (const nco_event* const*)seqs; in.seq_len = len; in.seq_strand = strand;
    in.model[0] = m0; in.model[1] = m1;
    in.default_p_stay = default_p_stay; in.default_p_skip = default_p_skip; in.train_drift = train_drift;
    nco_pm_params cp = {crt_pm[0], crt_pm[1], crt_pm[2], crt_pm[3], crt_pm[4], crt_pm[5]}, np = cp;
    nco_st_params cs[2] = {{crt_st[0], crt_st[1]}, {crt_st[2], crt_st[3]}}, ns[2] = {cs[0], cs[1]};
    float fit = nco_train_one_round(&in, &cp, cs, &np, ns, done, train_scaling, train_transitions);
    new_pm[0] = np.scale; new_pm[1] = np.shift; new_pm[2] = np.drift;
    new_pm[3] = np.var; new_pm[4] = np.scale_sd; new_pm[5] = np.var_sd;
    new_st[0] = ns[0].p_stay; new_st[1] = ns[0].p_skip; new_st[2] = ns[1].p_stay; new_st[3] = ns[1].p_skip;
    for (size_t k = 0; k < n_seqs; ++k) free(seqs[k]);
    free(seqs); free(len); free(m0); free(m1);
    return fit;
}


/* ------------------------------------------------------------------------------------------
 * Fast5_Summary  (src/nanocall/Fast5_Summary.hpp)  -- see the header for the parity status
 * ---------------------------------------------------------------------------------------- */

/* alg::mean_stdv_of<float> (hpptools alg.hpp, ABSENT) as used at Fast5_Summary.hpp:225-230,256-258 and
 * nanocall.cpp:633-635.  Same restatement as model_update_statistics above. */
void nco_mean_stdv(const float* v, size_t n, float* mean, float* stdv)
{
    float s = 0, s2 = 0;
    for (size_t i = 0; i < n; ++i) {
        s += v[i];
        s2 += v[i] * v[i];
    }
    float fn = (float)n;
    *mean = n > 0 ? s / fn : 0;
    float var = n > 1 ? (s2 - s * *mean) / (fn - 1) : 0;
    *stdv = var > 0 ? sqrtf(var) : 0;
}

static int cmp_float(const void* a, const void* b)
{
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* detect_abasic_level, Fast5_Summary.hpp:528-543 */
static float f5_detect_abasic_level(const nco_f5_opts* o, const nco_ed_event* ed, unsigned n)
{
    float* s = (float*)malloc(sizeof(float) * (n ? n : 1));
    for (unsigned i = 0; i < n; ++i) s[i] = (float)ed[i].mean;                 /* :539 */
    qsort(s, n, sizeof(float), cmp_float);                                     /* :541 */
    size_t idx = (size_t)((double)n * (1.0 - o->abasic_level_top_percent / 100.0));   /* :542 */
    float r = (float)((double)s[idx] + o->abasic_level_top_offset);            /* float + double -> double -> Float_Type */
    free(s);
    return r;
}

typedef struct { unsigned first, second; } f5_island;

/* detect_strands, Fast5_Summary.hpp:653-731 with find_islands_5_consec :545-571 */
static void f5_detect_strands(const nco_f5_opts* o, const nco_ed_event* ed, unsigned n, float abasic_level, unsigned sb[4])
{
    f5_island* isl = (f5_island*)malloc(sizeof(f5_island) * (n / 5 + 2));
    unsigned n_isl = 0;
    unsigned i = 0;
    while (i < n) {                                                           /* :552-569 */
        if (ed[i].mean >= abasic_level) {
            unsigned j = i + 1;
            while (j < n && ed[j].mean >= abasic_level) ++j;
            if (j - i >= 5) { isl[n_isl].first = i; isl[n_isl].second = j; ++n_isl; }
            i = j + 1;
        } else {
            ++i;
        }
    }
    /* merge islands within max(hp_start, hp_end) of each other, restarting after every merge (:665-676) */
    const unsigned gap = o->trim_margins[2] > o->trim_margins[3] ? o->trim_margins[2] : o->trim_margins[3];
    for (unsigned k = 1; k < n_isl; ++k) {
        if (isl[k - 1].second + gap >= isl[k].first) {
            isl[k - 1].second = isl[k].second;
            memmove(&isl[k], &isl[k + 1], sizeof(f5_island) * (n_isl - k - 1));
            --n_isl;
            k = 0;
        }
    }
    if (n_isl == 0) { free(isl); return; }                                    /* template only, :685-690 */
    /* island closest to the middle (alg::min_of = first minimum), :694-698 */
    unsigned best = 0, best_d = 0;
    for (unsigned k = 0; k < n_isl; ++k) {
        long a = labs((long)isl[k].first - (long)n / 2), b = labs((long)isl[k].second - (long)n / 2);
        unsigned d = (unsigned)(a < b ? a : b);
        if (k == 0 || d < best_d) { best = k; best_d = d; }
    }
    if (best_d > n / 6) { free(isl); return; }                                /* not in the middle third: template only, :700-713 */
    sb[0] = o->trim_margins[0];                                               /* :718-729 */
    if (isl[0].first < o->trim_margins[0] + o->trim_margins[2]) sb[0] = sb[0] > isl[0].second ? sb[0] : isl[0].second;
    sb[1] = isl[best].first - o->trim_margins[2];
    sb[2] = isl[best].first + o->trim_margins[3];
    sb[3] = n - o->trim_margins[1];
    if (isl[n_isl - 1].second > n - (o->trim_margins[3] + o->trim_margins[1]))
        sb[3] = sb[3] < isl[n_isl - 1].first ? sb[3] : isl[n_isl - 1].first;
    free(isl);
}

/* filter_ed_event, Fast5_Summary.hpp:734-745 */
static int f5_filter(const nco_ed_event* e, float abasic_level)
{
    if (e->mean >= abasic_level) return 0;
    if (e->stdv > 4.0) return 0;
    return 1;
}

size_t nco_f5_load_events(const nco_f5_summary* s, const nco_ed_event* ed, float sampling_rate, unsigned st, float* mean,
                          float* stdv, float* start, float* length)
{
    size_t n = 0;
    if (s->num_ed_events == 0) return 0;
    const unsigned ref = s->strand_bounds[s->scale_strands_together ? 0 : 2 * st];          /* :359 */
    for (unsigned j = s->strand_bounds[2 * st]; j < s->strand_bounds[2 * st + 1]; ++j) {   /* :351-364 */
        if (!f5_filter(&ed[j], s->abasic_level)) continue;
        nco_event e;
        nco_event_init(&e, (float)ed[j].mean, (float)ed[j].stdv,
                       (float)(ed[j].start - ed[ref].start) / sampling_rate,                /* long long -> float, float divide */
                       (float)ed[j].length / sampling_rate);
        mean[n] = e.mean; stdv[n] = e.stdv; start[n] = e.start; length[n] = e.length;
        ++n;
    }
    return n;
}

void nco_f5_summarize(const nco_f5_opts* o, const nco_ed_event* ed, size_t n_ed_file, float sampling_rate, int sst,
                      nco_f5_summary* out)
{
    memset(out, 0, sizeof(*out));
    if (sampling_rate < 1000.0 || sampling_rate > 10000.0) return;                           /* :168-172 */
    unsigned n = n_ed_file > o->max_ed_events ? o->max_ed_events : (unsigned)n_ed_file;      /* load_ed_events :505-525 */
    if (n < o->trim_margins[0] + o->trim_margins[1] + o->min_ed_events) return;             /* :185-191 */
    out->abasic_level = f5_detect_abasic_level(o, ed, n);                                   /* :193 */
    if (out->abasic_level <= 1.0) return;                                                    /* :194-200 */
    unsigned sb[4] = {o->trim_margins[0], n - o->trim_margins[1], 0, 0};                     /* :202 */
    if (!o->template_only) f5_detect_strands(o, ed, n, out->abasic_level, sb);
    memcpy(out->strand_bounds, sb, sizeof(sb));
    if (sb[1] <= sb[0]) return;                                                              /* :204-209 (bounds stay as detected) */
    out->num_ed_events = n;
    out->scale_strands_together = sst && sb[1] - sb[0] >= o->min_ed_events && sb[3] - sb[2] >= o->min_ed_events;   /* :210-212 */
    /* time lengths, :214-219 */
    for (unsigned st = 0; st < 2; ++st) {
        const unsigned cap = sb[2 * st + 1] > sb[2 * st] ? sb[2 * st + 1] - sb[2 * st] : 0;
        if (cap == 0) continue;
        float* buf = (float*)malloc(sizeof(float) * 4 * cap);
        size_t m = nco_f5_load_events(out, ed, sampling_rate, st, buf, buf + cap, buf + 2 * cap, buf + 3 * cap);
        if (m >= o->min_ed_events) out->time_length[st] = buf[2 * cap + m - 1] + buf[3 * cap + m - 1];
        free(buf);
    }
}

void nco_f5_initial_scaling(int together, const float r0[2], const float r1[2], const float m0[2], const float m1[2], float out[2])
{
    if (together) {                                                            /* Fast5_Summary.hpp:237-241 */
        float scale = (r0[1] / m0[1] + r1[1] / m1[1]) / 2;
        float shift = (r0[0] - scale * m0[0] + r1[0] - scale * m1[0]) / 2;
        out[0] = scale; out[1] = shift;
    } else {                                                                   /* :265-267 */
        float scale = r0[1] / m0[1];
        float shift = r0[0] - scale * m0[0];
        out[0] = scale; out[1] = shift;
    }
}


size_t nco_logf_mismatches(const float* x, const float* y, size_t n, long long* first_bad)
{
    size_t bad = 0;
    long long first = -1;
    for (size_t i = 0; i < n; ++i) {
        const float h = logf(x[i]);
        uint32_t a, b;
        memcpy(&a, &h, 4); memcpy(&b, &y[i], 4);
        if (a == b) continue;
        if (h != h && y[i] != y[i]) continue;   /* both NaN */
        if (first < 0) first = (long long)i;
        ++bad;
    }
    if (first_bad) *first_bad = first;
    return bad;
}
