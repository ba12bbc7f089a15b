// sr_flow.h -- counters of the tile-flow Cholesky (round 6; sr_flow.hip, the resident diagonal-block kernel of sr_factor.hip,
// launch plan in sr_capi_update.hip).  Unsigned words in device memory:
//   [SR_FLOW_GO]            epoch of the run whose Gram matrix is in place (published by the worker that takes task 0)
//   [SR_FLOW_ALIVE + d]     epoch of the run whose diagonal-block workgroup of output d is resident
//     -- these are never reset (zero at allocation, epochs only grow) --
//   [SR_FLOW_STATUS]        != 0: a wait ran into its time-out, everybody leaves; the host repeats the update by launches
//   [SR_FLOW_TASK]          next task
//   per output d, at SR_FLOW_HDR + d * sr_flow_words(nb):
//       dd[nb]              1: diagonal block factored and inverted
//       ac[nb][2 nb]        per block row and column of 64: row halves whose left-looking update is complete (2 = both)
//       tr[nb][2 nb]        the same for the block-row solve: the factor's block row at these 64 columns is final
//     -- zeroed in front of every run (one memset from SR_FLOW_STATUS on) --
#pragma once
#include "sr_common.h"

#define SR_FLOW_GO 0
#define SR_FLOW_ALIVE 1
#define SR_FLOW_STATUS 16
#define SR_FLOW_TASK 17
#define SR_FLOW_HDR 32

static inline __host__ __device__ long sr_flow_words(int nb) { return (long)nb + 4L * nb * nb; }

// tasks of one block row i of one output (band = blocks right of the diagonal block that go in 64 x 64 tiles):
//   left-looking updates (i >= 1): 3 tiles of the diagonal block, 4 per near block, 1 per far block
//   block-row solves:              4 per near block, 1 per far block
static inline __host__ __device__ int sr_flow_near(int nb, int i, int band) { const int rem = nb - 1 - i; return rem < band ? rem : band; }
static inline __host__ __device__ int sr_flow_nacc(int nb, int i, int band) {
    if (i == 0) return 0;
    const int nr = sr_flow_near(nb, i, band);
    return 3 + 4 * nr + (nb - 1 - i - nr);
}
static inline __host__ __device__ int sr_flow_ntr(int nb, int i, int band) {
    const int nr = sr_flow_near(nb, i, band);
    return 4 * nr + (nb - 1 - i - nr);
}

struct sr_flow_params {
    double *U, *W, *Wt;          // per output: U, W sU doubles apart, Wt sWt
    long sU, sWt;
    int Np, nb, n_out, band;
    long total;                  // tasks per output
    unsigned* flags;
    unsigned epoch;
    unsigned long long timeout;  // ticks of the 100 MHz clock a wait may last
    int acq;                     // agent-scope acquire behind every wait (1) or none (0: see sr_flow.hip)
};

int sr_launch_flow_workers(const sr_flow_params& p, int wgs, hipStream_t s);
int sr_launch_flow_diag_server(double* A, long lda, double* Wt, double* W, long ldw, int nb, int* info_dev, unsigned* flags,
                               unsigned epoch, double timeout_go_s, double timeout_s, hipStream_t s, const sr_batch* bt);
