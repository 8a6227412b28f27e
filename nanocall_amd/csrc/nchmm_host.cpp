// nchmm_host.cpp -- host-side prep of the HMM tables (pure CPU; compiled into libnanocall_hip.so).
//
// These are the pieces of the reference's hot path that must stay on the host because they go
// through libm (std::log / pow) and the called k-mer path has to be bit-exact: device logf is not
// glibc logf.  Each function names the reference code it reproduces; nothing here is shared with
// oracle/ (the oracle is an independent C restatement used only by the tests).
//
// Build flags matter: -ffp-contract=off (no FMA fusion), no -ffast-math.
#include "nanocall_hip.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "nchmm_internal.hpp"
#include "nchmm_kmer.hpp"

namespace {

enum Field {  // order of Pore_Model_State, src/nanocall/Pore_Model.hpp:85-96
    F_LEVEL_MEAN, F_LEVEL_STDV, F_SD_MEAN, F_SD_STDV, F_SD_LAMBDA,
    F_LOG_LEVEL_MEAN, F_LOG_LEVEL_STDV, F_LOG_SD_MEAN, F_LOG_SD_STDV, F_LOG_SD_LAMBDA, F_COUNT
};

}  // namespace

extern "C" {

const char* nchmm_strerror(int code)
{
    switch (code) {
    case NCHMM_OK: return "ok";
    case NCHMM_E_INVALID: return "invalid argument";
    case NCHMM_E_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case NCHMM_E_HIP: return "HIP runtime error";
    case NCHMM_E_TOPOLOGY: return "transitions are not the stay/step/skip-1 graph of compute_transitions_fast";
    case NCHMM_E_NOMEM: return "out of memory";
    case NCHMM_E_NUMERIC: return "read decoded to -INF/NaN in every state";
    case -7: return "I/O error (HDF5 / FAST5; see nchmm_fast5_last_error)";
    default: return "unknown error";
    }
}

int nchmm_abi_version(void) { return 2; }

// Pore_Model::load_from_vector, Pore_Model.hpp:220-239 (+ update_sd_lambda :112, update_logs :118-124)
int nchmm_model_load(const float* table, float* state)
{
    if (!table || !state) return NCHMM_E_INVALID;
    for (unsigned i = 0; i < NCHMM_N_STATES; ++i) {
        float* s = state + (size_t)i * F_COUNT;
        s[F_LEVEL_MEAN] = table[4 * i + 0];
        s[F_LEVEL_STDV] = table[4 * i + 1];
        s[F_SD_MEAN] = table[4 * i + 2];
        s[F_SD_STDV] = table[4 * i + 3];
        // pow() on (float, double) promotes to double; the quotient is rounded once to float
        s[F_SD_LAMBDA] = static_cast<float>(std::pow(static_cast<double>(s[F_SD_MEAN]), 3.0)
                                            / std::pow(static_cast<double>(s[F_SD_STDV]), 2.0));
        s[F_LOG_LEVEL_MEAN] = std::log(s[F_LEVEL_MEAN]);
        s[F_LOG_LEVEL_STDV] = std::log(s[F_LEVEL_STDV]);
        s[F_LOG_SD_MEAN] = std::log(s[F_SD_MEAN]);
        s[F_LOG_SD_STDV] = 0.0f;
        s[F_LOG_SD_LAMBDA] = std::log(s[F_SD_LAMBDA]);
    }
    return NCHMM_OK;
}

// Pore_Model::scale :190-201 and Pore_Model_State::scale :126-138
int nchmm_model_scale(float* state, const float params[6])
{
    if (!state || !params) return NCHMM_E_INVALID;
    const float scale = params[0], shift = params[1], var = params[3], scale_sd = params[4], var_sd = params[5];
    const float log_var = std::log(var), log_scale_sd = std::log(scale_sd), log_var_sd = std::log(var_sd);
    for (unsigned i = 0; i < NCHMM_N_STATES; ++i) {
        float* s = state + (size_t)i * F_COUNT;
        s[F_LEVEL_MEAN] = s[F_LEVEL_MEAN] * scale + shift;
        s[F_LEVEL_STDV] = s[F_LEVEL_STDV] * var;
        s[F_SD_MEAN] = s[F_SD_MEAN] * scale_sd;
        s[F_SD_LAMBDA] = s[F_SD_LAMBDA] * var_sd;
        s[F_SD_STDV] = static_cast<float>(std::pow(std::pow(static_cast<double>(s[F_SD_MEAN]), 3.0)
                                                   / static_cast<double>(s[F_SD_LAMBDA]), .5));
        s[F_LOG_LEVEL_MEAN] = std::log(s[F_LEVEL_MEAN]);
        s[F_LOG_LEVEL_STDV] += log_var;
        s[F_LOG_SD_MEAN] += log_scale_sd;
        s[F_LOG_SD_LAMBDA] += log_var_sd;
    }
    return NCHMM_OK;
}

int nchmm_model_pack6(const float* state, float* t6)
{
    if (!state || !t6) return NCHMM_E_INVALID;
    for (unsigned i = 0; i < NCHMM_N_STATES; ++i) {
        const float* s = state + (size_t)i * F_COUNT;
        float* o = t6 + (size_t)i * 6;
        o[0] = s[F_LEVEL_MEAN]; o[1] = s[F_LEVEL_STDV]; o[2] = s[F_LOG_LEVEL_STDV];
        o[3] = s[F_SD_MEAN]; o[4] = s[F_SD_LAMBDA]; o[5] = s[F_LOG_SD_LAMBDA];
    }
    return NCHMM_OK;
}

// State_Transitions::compute_transitions_fast :181-224 + update_fields :79-104
int nchmm_transitions_fast(float p_skip, float p_stay, uint32_t* row_ptr, uint16_t* pred, float* logw,
                           uint32_t* n_arcs)
{
    if (!row_ptr || !pred || !logw) return NCHMM_E_INVALID;
    using nchmm::Kmer6;
    float p_step, p_skip_1;
    nchmm::step_params(p_skip, p_stay, p_step, p_skip_1);
    auto trans_prob = [&](unsigned i, unsigned j) { return nchmm::trans_prob(i, j, p_stay, p_step, p_skip_1); };
    // destination-major build: the predecessors of j are j, (x<<10)|(j>>2), (xy<<8)|(j>>4)
    uint32_t n = 0;
    for (unsigned j = 0; j < NCHMM_N_STATES; ++j) {
        unsigned cand[21]; unsigned m = 0;
        cand[m++] = j;
        for (unsigned x = 0; x < 4; ++x) cand[m++] = (x << 10) | (j >> 2);
        for (unsigned xy = 0; xy < 16; ++xy) cand[m++] = (xy << 8) | (j >> 4);
        std::sort(cand, cand + m);
        m = static_cast<unsigned>(std::unique(cand, cand + m) - cand);
        row_ptr[j] = n;
        for (unsigned k = 0; k < m; ++k) {
            pred[n] = static_cast<uint16_t>(cand[k]);
            logw[n] = std::log(trans_prob(cand[k], j));  // std::log(float) :216
            ++n;
        }
    }
    row_ptr[NCHMM_N_STATES] = n;
    if (n_arcs) *n_arcs = n;
    return NCHMM_OK;
}

// Event::update_logs :35-45 and Event_Sequence::apply_drift_correction :77-84
int nchmm_events_prepare(size_t n, const float* mean, float* stdv, const float* start, float drift,
                         float* corrected_mean, float* log_stdv)
{
    if (n && (!mean || !stdv || !corrected_mean || !log_stdv)) return NCHMM_E_INVALID;
    if (n && drift != 0.0f && !start) return NCHMM_E_INVALID;
    auto body = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            if (stdv[i] == 0.0) stdv[i] = static_cast<float>(0.01);
            log_stdv[i] = std::log(stdv[i]);
            float cm = mean[i];
            if (start) cm -= drift * start[i];
            corrected_mean[i] = cm;
        }
    };
    // element-wise: big batches (millions of events, ~5 ns of libm each) are spread over the host cores
    if (n >= (size_t)1 << 16) nchmm::parallel_for((n + 16383) / 16384, [&](size_t lo, size_t hi) { body(lo * 16384, std::min(n, hi * 16384)); });
    else body(0, n);
    return NCHMM_OK;
}

// Viterbi::fill_move_seq :144-150, Event_Sequence::get_base_seq Event.hpp:85-99
int nchmm_base_seq(size_t n, const uint16_t* state, int32_t* move, char* seq, size_t* seq_len)
{
    using nchmm::Kmer6;
    if (n && (!state || !seq)) return NCHMM_E_INVALID;
    size_t len = 0;
    for (size_t i = 0; i < n; ++i) {
        if (state[i] >= NCHMM_N_STATES) return NCHMM_E_INVALID;
        unsigned mv = i > 0 ? Kmer6::min_skip(state[i - 1], state[i]) : 0u;
        if (move) move[i] = static_cast<int32_t>(mv);
        char km[6];
        Kmer6::to_chars(state[i], km);
        if (i == 0) { std::memcpy(seq, km, 6); len = 6; continue; }
        unsigned a = std::min(mv, 6u), b = 6 - a;
        std::memcpy(seq + len, km + b, a);
        len += a;
    }
    if (seq) seq[len] = 0;
    if (seq_len) *seq_len = len;
    return NCHMM_OK;
}

// write_fasta, nanocall.cpp:584-591
int nchmm_write_fasta(const char* name, const char* seq, unsigned line_width, char* out, size_t cap,
                      size_t* written)
{
    if (!name || !seq || !out || line_width == 0) return NCHMM_E_INVALID;
    const size_t L = std::strlen(seq), nl = std::strlen(name);
    const size_t need = 1 + nl + 1 + L + (L + line_width - 1) / line_width + 1;
    if (cap < need) return NCHMM_E_NOMEM;
    size_t n = 0;
    out[n++] = '>'; std::memcpy(out + n, name, nl); n += nl; out[n++] = '\n';
    for (size_t pos = 0; pos < L; pos += line_width) {
        size_t c = std::min<size_t>(line_width, L - pos);
        std::memcpy(out + n, seq + pos, c); n += c; out[n++] = '\n';
    }
    out[n] = 0;
    if (written) *written = n;
    return NCHMM_OK;
}

// Parameter_Trainer::init, Parameter_Trainer.hpp:30-57
int nchmm_st_train_kmers(uint16_t* out, uint32_t* count)
{
    using nchmm::Kmer6;
    if (!out || !count) return NCHMM_E_INVALID;
    uint32_t n = 0;
    for (unsigned i = 0; i < NCHMM_N_STATES; ++i) {
        if (Kmer6::max_self_overlap(i) > 0) continue;
        bool all_good = true;
        for (unsigned b1 = 0; b1 < 4 && all_good; ++b1)
            if (Kmer6::max_self_overlap((Kmer6::suffix(i, 5) << 2) + b1) > 1) all_good = false;
        if (all_good) out[n++] = static_cast<uint16_t>(i);
    }
    *count = n;
    return NCHMM_OK;
}

// Parameter_Trainer::train_pm_params, Parameter_Trainer.hpp:297-427, given the per-event inner
// sums {s0,s1,s2,l0,l1,l2} (:273-296, produced on the GPU by nchmm_fwbw).  Outer accumulation in
// double, products in float, 3x3 solve with scaled partial pivoting -- operation for operation.
int nchmm_train_pm_finish(size_t n_events, const float* pm_sums, const float* mean, const float* stdv,
                          const float* start, int train_drift, const float crt_pm[6], float new_pm[6], int* done)
{
    if (!pm_sums || !mean || !stdv || !crt_pm || !new_pm || !done || (train_drift && !start)) return NCHMM_E_INVALID;
    // acc = {A00, A01, A11, B0, B1, A02, A12, A22, B2, D, V_numer, V_denom, U_pos}
    double acc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < n_events; ++i) {
        const float* s = pm_sums + 6 * i;
        const float* l = s + 3;
        const float x_i = mean[i], y_i = stdv[i], t_i = start ? start[i] : 0.0f;
        acc[0] += s[0];
        acc[1] += s[1];
        acc[2] += s[2];
        acc[3] += s[0] * x_i;
        acc[4] += s[1] * x_i;
        if (train_drift) {
            acc[5] += s[0] * t_i;
            acc[6] += s[1] * t_i;
            acc[7] += s[0] * t_i * t_i;
            acc[8] += s[0] * x_i * t_i;
        }
        acc[9] += s[0] * x_i * x_i;
        acc[10] += l[2] * y_i;
        acc[11] += l[1];
        acc[12] += l[0] / y_i;
    }
    return nchmm_train_pm_solve(n_events, acc, train_drift, crt_pm, new_pm, done);
}

// The solve half of train_pm_params (Parameter_Trainer.hpp:314-427) from the thirteen outer sums
// {A00, A01, A11, B0, B1, A02, A12, A22, B2, D, V_numer, V_denom, U_pos} (nchmm_train_pm_finish above, or the
// device reduction of nchmm_em_round).
int nchmm_train_pm_solve(size_t n_events, const double acc[13], int train_drift, const float crt_pm[6], float new_pm[6], int* done)
{
    if (!acc || !crt_pm || !new_pm || !done) return NCHMM_E_INVALID;
    *done = 0;
    double A[3][3] = {{acc[0], acc[1], acc[5]}, {0, acc[2], acc[6]}, {0, 0, acc[7]}}, B[3] = {acc[3], acc[4], acc[8]};
    const double D = acc[9], V_numer = acc[10], V_denom = acc[11], U_pos = acc[12];
    A[1][0] = A[0][1]; A[2][0] = A[0][2]; A[2][1] = A[1][2];
    if (!train_drift) A[2][2] = 1.0;
    double Ac[3][3], Bc[3], C[3];
    std::memcpy(Ac, A, sizeof(A)); std::memcpy(Bc, B, sizeof(B));
    for (unsigned i = 0; i < 3; ++i) C[i] = std::max(A[i][0], std::max(A[i][1], A[i][2]));   // alg::max_value_of :328
    for (unsigned i = 0; i < 3; ++i) {   // :340-386
        unsigned p = i;
        double p_val = std::abs(A[i][i]) / C[p];
        for (unsigned i2 = i + 1; i2 < 3; ++i2) {
            const double v = std::abs(A[i2][i]) / C[i2];
            if (v > p_val) { p = i2; p_val = v; }
        }
        if (p_val < 1e-7) {   // singular: keep the current parameters (:355-360)
            *done = 1;
            std::memcpy(new_pm, crt_pm, 6 * sizeof(float));
            return NCHMM_OK;
        }
        if (p > i) { std::swap(A[i], A[p]); std::swap(B[i], B[p]); std::swap(C[i], C[p]); }
        for (p = i + 1; p < 3; ++p) {
            const double m = A[p][i] / A[i][i];
            A[p][i] = 0;
            for (unsigned j = i + 1; j < 3; ++j) A[p][j] -= m * A[i][j];
            B[p] -= m * B[i];
        }
    }
    // each unknown is stored as float before the next one reads it (:388-390)
    const float c_hat = static_cast<float>(B[2] / A[2][2]);
    const float b_hat = static_cast<float>((B[1] - A[1][2] * c_hat) / A[1][1]);
    const float a_hat = static_cast<float>((B[0] - A[0][1] * b_hat - A[0][2] * c_hat) / A[0][0]);
    const double d_numer = (D + a_hat * a_hat * Ac[0][0] + b_hat * b_hat * Ac[1][1] + c_hat * c_hat * Ac[2][2]
                            + 2.0 * a_hat * b_hat * Ac[0][1] + 2.0 * a_hat * c_hat * Ac[0][2]
                            + 2.0 * b_hat * c_hat * Ac[1][2] - 2.0 * (a_hat * Bc[0] + b_hat * Bc[1] + c_hat * Bc[2]));
    const float d_hat = static_cast<float>(std::sqrt(d_numer / static_cast<double>(n_events)));        // :417
    const float v_hat = static_cast<float>(V_numer / V_denom);                                          // :422
    const float u_hat = static_cast<float>(static_cast<double>(n_events) / (U_pos - V_denom / v_hat));  // :426
    new_pm[0] = b_hat; new_pm[1] = a_hat; new_pm[2] = c_hat; new_pm[3] = d_hat; new_pm[4] = v_hat; new_pm[5] = u_hat;
    return NCHMM_OK;
}

// Parameter_Trainer::train_st_params, Parameter_Trainer.hpp:516-530: combine the per-window log-sums
// {denom, stay_num, skip_num} of one strand into (p_stay, p_skip), with the [.05, .4] reset.
int nchmm_train_st_finish(size_t n_win, const float* st_sums, float* p_stay, float* p_skip)
{
    if ((n_win && !st_sums) || !p_stay || !p_skip) return NCHMM_E_INVALID;
    double acc[3] = {0, 0, 0};
    for (size_t w = 0; w < n_win; ++w)
        for (int k = 0; k < 3; ++k) acc[k] += std::exp(static_cast<double>(st_sums[3 * w + k]));
    float ps = static_cast<float>(acc[1] / acc[0]), pk = static_cast<float>(acc[2] / acc[0]);   // NaN for an empty strand, as in the reference
    if (ps < .05 || ps > .4 || pk < .05 || pk > .4) {
        ps = std::min(std::max(ps, .05f), .4f);
        pk = std::min(std::max(pk, .05f), .4f);
    }
    *p_stay = ps; *p_skip = pk;
    return NCHMM_OK;
}

}  // extern "C"
