/*
 * lpx_test.h -- TEST-ONLY entry of liblpx.so.  NOT part of the drop-in boundary (include/lpx.h): a host application never
 * includes this file and a C# shim never mirrors it.
 *
 * The CPU-only test-suite drives the sharded host logic (frontier partition, per-level all-reduce, rebalancing,
 * termination) under torch.distributed/gloo with world_size 2 on a box without a GPU.  It does so by standing in for
 * the device loops with the two callbacks below; they are installed per calling thread, apply to the lpx_solve calls
 * that follow, and are NULL in every product path (nothing in the package installs them).
 *   node_lp      replaces lpx_multi_run + lpx_tableau_solution for ONE prepared node tableau
 *                (T is R x C row-major, modified in place; returns the LPX_* status)
 *   knap_relax   replaces lpx_knapsack_relax_batch (same argument meaning)
 * A negative return value of node_lp is a device failure: the search throws, as it does when lpx_multi_run fails.
 */
#ifndef LPX_TEST_H
#define LPX_TEST_H

#include "lpx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lpx_test_seams {
    int (*node_lp)(void* user, double* T, int R, int C, int32_t* basis, int dual, int repaired,
                   int max_iter, int nvars, double* x, double* z, int64_t* pivots);
    int (*knap_relax)(void* user, int count, const int32_t* off, const int32_t* fix_idx,
                      const int8_t* fix_val, double* profit, double* weight, int32_t* frac_idx,
                      double* frac_val);
    void* user;
    /* > 0: the branch-and-bound searches of this thread throw LPX_ENOMEM when a group of node LPs is about to be solved
     * after this many nodes were handed out (real device loops or the stand-in above): the peer-failure path of the
     * sharded searches -- every rank must come back with an error, none may wait in a collective for good. */
    int64_t fail_after_nodes;
} lpx_test_seams;

void lpx_test_set_seams(const lpx_test_seams* seams);   /* NULL = none */

/* The id hand-over of lpx_comm_init_tcp (include/lpx.h) without a device or RCCL behind it: rank 0 serves id[128] on
 * host:port to the world - 1 other ranks, which receive it into id.  Returns 0 or LPX_EDEVICE (lpx_last_error). */
int lpx_test_comm_exchange_id(int rank, int world, const char* host, int port, uint8_t* id);

#ifdef __cplusplus
}
#endif
#endif /* LPX_TEST_H */
