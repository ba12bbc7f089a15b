// sr_flow.h -- counters of the tile-flow Cholesky (round 6; sr_flow.hip, the resident diagonal-block kernel of sr_factor.hip,
// launch plan in sr_capi_update.hip).  Unsigned words in device memory:
//   [SR_FLOW_GO]            epoch of the run whose Gram matrix is in place (published by the worker that takes task 0)
//   [SR_FLOW_ALIVE + d]     epoch of the run whose diagonal-block workgroup of output d is resident
//     -- these are never reset (zero at allocation, epochs only grow) --
//   [SR_FLOW_STATUS]        != 0: a wait ran into its time-out, everybody leaves; the host repeats the update by launches
//   [SR_FLOW_TASK + 3]      next task
//   per output d, at SR_FLOW_HDR + d * sr_flow_words(nb):
//       dd[nb]              1: diagonal block factored and inverted
//       ac[nb][2 nb]        per block row and column of 64: row halves whose left-looking update is complete (2 = both)
//       tr[nb][2 nb]        the same for the block-row solve: the factor's block row at these 64 columns is final
//       ap[nb][nb]          per 128 x 128 block: panels of factor rows it has taken (sr_flow_seg below)
//       tk[nb]              diagnostics: 100 MHz ticks from the go to dd[kb] = 1
//     -- zeroed in front of every run (one memset from SR_FLOW_STATUS on) --
#pragma once
#include "sr_common.h"

#define SR_FLOW_GO 0
#define SR_FLOW_ALIVE 1
#define SR_FLOW_STATUS 16
#define SR_FLOW_TASK 17
#define SR_FLOW_STATS 24       /* 6 task kinds x [count, ticks in the task, ticks of it spent waiting, -] (100 MHz), diagnostics */
#define SR_FLOW_HDR 64

static inline __host__ __device__ long sr_flow_words(int nb) { return 2L * nb + 9L * nb * nb; }

// Task plan (sr_flow.hip has the picture).  Block rows go in PANELS of `panel` blocks; behind a panel every 128 x 128 block
// takes the panel's rows in ONE product (UPD; ap[i][j] counts the panels block (i, j) has taken) -- except the `band` of the
// NEXT panel's rows, whose 64 x 64 tiles take them left-looking.  Three kinds of tasks, each enumerated in its own order:
//   critical: per block row i -- 4 solves per near block of row i, then [3 diagonal tiles, 4 updates per near block] of row i + 1
//   far:      per block row i -- one task per far block (update by the panel's rows above + solve; first row of a panel: solve)
//   updates:  panel by panel, the blocks behind the panel in row-major order (the next panel's rows first)
// ONE order for all of them: per block row [its share of the updates by the previous panel -- with a panel's first
// row all those the panel's own rows need --] [solves of the row] [far blocks of the row] [updates of the next row's band],
// handed out by one fetch-and-add; a task that cannot start yet waits inside.  Everything a task waits for is in front of
// it in this order (tests/test_host_logic.py checks the plan against exactly that).
struct sr_flow_seg { int start_c, start_f, start_b, start_m; };      // per block row (and one behind the last): first task of
                                                                     // each queue / of the one order

static inline __host__ __device__ int sr_flow_near(int nb, int i, int band) { const int rem = nb - 1 - i; return rem < band ? rem : band; }
struct sr_flow_params {
    double *U, *W, *Wt;          // per output: U, W sU doubles apart, Wt sWt
    long sU, sWt;
    int Np, nb, n_out, band, panel;
    long total, total_far, total_upd;   // tasks per output: critical, far blocks, panel updates
    long total_m;                // tasks per output in the one order
    int prio;                    // issue priorities for far blocks / next-panel updates (1) or for the band only (0)
    int keep, exit_row;          // workgroups with blockIdx >= keep leave after a row task of block row >= exit_row
    const sr_flow_seg* segs;     // nb + 1 segments
    unsigned* flags;
    unsigned epoch;
    unsigned long long timeout;  // ticks of the 100 MHz clock a wait may last
    int acq;                     // agent-scope acquire behind EVERY wait (lab switch; 0: only where sr_flow.hip says it is needed)
};

// host: the segments of (nb, band, panel); returns the critical tasks per output, *total_far / *total_upd the other queues'
long sr_flow_plan(int nb, int band, int panel, sr_flow_seg* segs, long* total_far, long* total_upd, long* total_m);
int sr_launch_flow_workers(const sr_flow_params& p, int wgs, hipStream_t s);
// one workgroup that ends when run `epoch` has every factor row above block row X final (X = nb: the whole factor)
int sr_launch_flow_gate(unsigned* flags, unsigned epoch, int n_out, int nb, int X, double timeout_s, hipStream_t s);
int sr_launch_flow_diag_server(double* A, long lda, double* Wt, double* W, long ldw, int nb, int panel, int* info_dev,
                               unsigned* flags, unsigned epoch, double timeout_go_s, double timeout_s, hipStream_t s,
                               const sr_batch* bt);
