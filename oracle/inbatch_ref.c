/*
 * inbatch_ref.c -- plain-C restatement of the in-batch debiased cross-entropy of westlake-repl/IDvs.MoRec
 * (inbatch_sasrec_e2e_text/model/model.py:32-33,45-67).  TEST INFRASTRUCTURE ONLY: it exists to check the HIP
 * kernels and the numpy oracle (oracle/morec_oracle); nothing in idvs.morec_amd links or calls it.
 * Pinned against the golden vectors captured from the imported reference (tests/test_oracle_c.py).
 *
 * Layout: B users, S = max_seq_len, slots = B*(S+1) local item slots (ids, left padded with 0),
 * columns = Nc pool slots (== local slots when not pooled), rows = B*S.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* model.py:45-48 -- label of row (i, j-1) is column i*(S+1)+j, j = 1..S (plus the pool offset of this rank) */
void morec_ref_labels(int B, int S, int col_offset, int64_t* out) {
    for (int i = 0; i < B; ++i)
        for (int j = 1; j <= S; ++j) out[i * S + (j - 1)] = (int64_t)col_offset + (int64_t)i * (S + 1) + j;
}

/* model.py:51-52 -- column valid iff cat(log_mask, ones)[slot] != 0 */
void morec_ref_column_valid(int B, int S, const float* log_mask, uint8_t* out) {
    for (int i = 0; i < B; ++i) {
        for (int j = 0; j < S; ++j) out[i * (S + 1) + j] = log_mask[i * S + j] != 0.0f;
        out[i * (S + 1) + S] = 1;
    }
}

/* model.py:54-63 -- masked[r][c] = column invalid, or id(c) is one of the row's user's S+1 ids and c is not the label */
void morec_ref_mask(int B, int S, int Nc, int col_offset, const int64_t* row_ids, const int64_t* col_ids,
                    const uint8_t* col_valid, uint8_t* masked /* [B*S][Nc] */) {
    for (int i = 0; i < B; ++i)
        for (int j = 0; j < S; ++j) {
            const int64_t label = (int64_t)col_offset + (int64_t)i * (S + 1) + j + 1;
            uint8_t* row = masked + ((size_t)i * S + j) * Nc;
            for (int c = 0; c < Nc; ++c) {
                int member = 0;
                for (int k = 0; k <= S; ++k) member |= (row_ids[i * (S + 1) + k] == col_ids[c]);
                row[c] = (uint8_t)(!col_valid[c] || (member && c != label));
            }
        }
}

/* model.py:33,49-50,65-67 -- sum over valid rows of (logsumexp(logits) - logits[label]); logits in fp32 like the
 * reference (matmul accumulates in double here, rounded once), masked cells = -1e4.  Returns the SUM and the count. */
double morec_ref_loss_sum(int B, int S, int D, int Nc, int col_offset, const float* P, const float* E,
                          const int64_t* row_ids, const int64_t* col_ids, const float* col_logpop,
                          const float* log_mask, const uint8_t* col_valid, int64_t* n_valid_out) {
    const int Nr = B * S;
    uint8_t* masked = (uint8_t*)malloc((size_t)Nr * Nc);
    float* logit = (float*)malloc(sizeof(float) * Nc);
    morec_ref_mask(B, S, Nc, col_offset, row_ids, col_ids, col_valid, masked);
    double total = 0.0;
    int64_t nv = 0;
    for (int r = 0; r < Nr; ++r) {
        if (log_mask[r] == 0.0f) continue;
        const int i = r / S, j = r % S;
        const int64_t label = (int64_t)col_offset + (int64_t)i * (S + 1) + j + 1;
        float mx = -INFINITY;
        for (int c = 0; c < Nc; ++c) {
            double acc = 0.0;
            for (int d = 0; d < D; ++d) acc += (double)P[(size_t)r * D + d] * (double)E[(size_t)c * D + d];
            float v = (float)acc - col_logpop[c];
            if (masked[(size_t)r * Nc + c]) v = -1e4f;
            logit[c] = v;
            if (v > mx) mx = v;
        }
        double se = 0.0;
        for (int c = 0; c < Nc; ++c) se += exp((double)logit[c] - (double)mx);
        total += (double)mx + log(se) - (double)logit[label];
        ++nv;
    }
    free(masked);
    free(logit);
    if (n_valid_out) *n_valid_out = nv;
    return total;
}
