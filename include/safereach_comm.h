/* safereach_comm.h -- C-ABI of libsafereach_comm.so (MI355X / gfx950): the ONE exchange step of the path.
 *
 * The hot path shards over queries; after a model update the posterior state (alpha, U^-1) travels once from the GPU
 * that factorised to the others (one RCCL broadcast over xGMI).  The reference has no multi-GPU path,
 * so nothing of it is replaced here.  The Python mirror replicates with torch.distributed (backend "nccl" = RCCL,
 * safe_exploration_amd/parallel.py); this library offers the same step to a plain C / C++ host that drives several GPUs
 * from one process.  It is a shared object of its own, linked against libsafereach.so and librccl.so, so that
 * libsafereach.so stays loadable next to PyTorch (which bundles its own RCCL).  Status codes and error conventions as in
 * safereach.h; sr_comm_last_error() holds the message of the last failing sr_comm_* call of the thread.
 */
#ifndef SAFEREACH_COMM_H
#define SAFEREACH_COMM_H
#include "safereach.h"

#ifdef __cplusplus
extern "C" {
#endif

/* one RCCL communicator (and one stream) per device, single process (ncclCommInitAll) */
int         sr_comm_init_all(int ndev, const int* devices);
/* handles[i] lives on devices[i]; all of the same shape (sr_gp_dims); handles[root] is factorised, the others have their
 * data set (sr_gp_set_data[_general]: Z, targets, hyper-parameters).  alpha and the PACKED upper triangle of U^-1 travel
 * (n_out * (N + N (N + 1) / 2) doubles, pieces of <= 64 MB through one staging buffer per device; sr_gp_export_packed /
 * sr_gp_import_packed of safereach.h): every handle is ready to predict, no copy of the factor's size is staged. */
int         sr_comm_bcast(sr_gp_t* handles, int ndev, int root);
int         sr_comm_destroy(void);
const char* sr_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SAFEREACH_COMM_H */
