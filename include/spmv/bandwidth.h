// spmv/bandwidth.h — the algorithmic-byte model behind every GB/s figure.
//
// Formulas follow the reference (src/bandwidth.cpp:22-88):
//   CSR bytes = nnz*(4+4) + (rows+1)*4 + cols*4 + rows*4
//   ELL bytes = rows*K*(4+4) + cols*4 + rows*4
// The peak is table-driven for HBM parts (gfx950: 8000 GB/s) instead of the
// reference's DDR clock*bus formula.
#ifndef SPMV_BANDWIDTH_H
#define SPMV_BANDWIDTH_H

#include "csr_matrix.h"
#include "ell_matrix.h"

namespace spmv {

struct BandwidthMetrics {
    float theoretical_bandwidth_gb_s;
    float achieved_bandwidth_gb_s;
    float efficiency;   // achieved / theoretical, capped at 1

    BandwidthMetrics() : theoretical_bandwidth_gb_s(0.0f),
                         achieved_bandwidth_gb_s(0.0f),
                         efficiency(0.0f) {}
};

BandwidthMetrics compute_bandwidth_csr(const CSRMatrix* A, float elapsed_ms);
BandwidthMetrics compute_bandwidth_ell(const ELLMatrix* A, float elapsed_ms);

float get_gpu_peak_bandwidth();   // GB/s of the current device's memory system

} // namespace spmv

#endif // SPMV_BANDWIDTH_H
