/* rapidnet_debug.h -- TEST HOOKS of librapidnet_hip.so.  Not part of the drop-in boundary (include/rapidnet.h): nothing here replaces a
 * member of the reference; the parity tests use these entry points the way the reference's own tests reach into protected members
 * (TestSmpcController.cu:134-159).  A production caller has no use for them. */
#ifndef RAPIDNET_DEBUG_H_
#define RAPIDNET_DEBUG_H_

#include "rapidnet.h"

#ifdef __cplusplus
extern "C" {
#endif

/* test hook for a caller's leak check (SmpcController.cu:1612-1623): the NEXT rn_control_action allocates `bytes` of device memory
 * that stay with the context */
int rn_debug_inject_allocation(rn_ctx *ctx, size_t bytes);

/* Test hooks for the sharded sweep: run rn_solve_step in two halves around the exchange -- phase 1 stops after the
 * cut parents' LOCAL children sums (solveSumChildren, Utilities.cu:168-201) are in the exchange buffer, phase 2 resumes from a
 * buffer the caller has summed over the shards (rn_debug_cut_buffer reads / writes it: cutParents * (nv + 2 nx) reals). */
int rn_debug_sweep_phase(rn_ctx *ctx, int phase);
int rn_debug_cut_buffer(rn_ctx *ctx, int write, double *host, size_t n);

/* An in-process stand-in for the communicator, so that the device-resident sharded path (checkpoint, dist^2 tail, verdict vote,
 * replay) can run with several ranks on a box where RCCL cannot (one GPU: "Duplicate GPU detected").
 * rn_debug_set_allreduce installs a callback that is called wherever the library would call ncclAllReduce(in place):
 * devBuf / count / isF64 / op describe the payload and the reduction, stream is the context's hipStream_t; it must return 0 after the
 * reduced values are (or are stream-ordered to be) in devBuf.  rn_debug_local_group_* is such a callback inside the library for
 * `nranks` contexts of ONE process, each driven by its own host thread: stream sync, barrier, every rank sums the payloads
 * in rank order (bitwise the same on all ranks), barrier.  A rank that does not arrive within 120 s
 * ($RAPIDNET_GROUP_TIMEOUT_S when the group is created) fails the others with RN_E_COMM instead of hanging them. */
typedef int (*rn_allreduce_fn)(void *user, void *devBuf, size_t count, int isF64, int op /* 0 = sum, 2 = max (ncclRedOp_t) */, void *stream);
int rn_debug_set_allreduce(rn_ctx *ctx, rn_allreduce_fn fn, void *user);
int rn_debug_local_group_create(int nranks, void **group);
int rn_debug_local_group_join(rn_ctx *ctx, void *group, int rank);
int rn_debug_local_group_destroy(void *group);

/* one-shot exchange: wires the inboxes (rn_peer_inbox_create) of `nranks` contexts of ONE process -- same address space, no IPC
 * handle needed -- contexts in rank order */
int rn_debug_peer_inbox_connect_local(rn_ctx **ctxs, int nranks);

/* one-shot exchange: sets the sequence number of the context's LAST exchange (the next one carries seq + 1; every rank must be given the
 * same value) -- lets a test walk the 32-bit tag across its wrap, where two values are skipped so that the parity which selects the
 * inbox buffer keeps alternating */
int rn_debug_peer_seq(rn_ctx *ctx, unsigned int seq);

/* test of the guard-mode detector itself (rn_guard_check): overwrites `nbytes` (1 .. 256) right behind the payload of the context's
 * first buffer */
int rn_debug_guard_poke(rn_ctx *ctx, int nbytes);

/* Launch-shape choices the library makes by problem size, forced from outside: the parity tests run the kernels that only large
 * trees take (three slabs per workgroup, double-buffered dual update, ...) on small ones, and the A/B tools time a shipped choice against
 * its alternative.  value = -1 gives the choice back to the library.  Before the factor step only: RN_E_STATE afterwards.  Every knob
 * selects between forms with IDENTICAL results (bitwise where the test of the knob says so); none changes what is computed. */
enum {
    RN_KNOB_DUAL_TRIPS = 0,        /* k_dual_stage: 16-byte vectors per thread (1 .. 64) */
    RN_KNOB_DUAL_PIPE = 1,         /* k_dual_stage: 1 = one vector at a time, 2 = double-buffered */
    RN_KNOB_VLV_WIDE = 2,          /* v / Lv products: slabs per workgroup of k_gemm_vlv_wide (2, 3; 0 / 1 = k_gemm_vlv) */
    RN_KNOB_SLAB_PIPE = 3,         /* slab products: 1 = the software-pipelined MFMA loop, 0 = the lean one */
    RN_KNOB_SLAB_FRAG = 4,         /* slab products: 0 = A operands from the column-major operators instead of the fragment-ordered copies */
    RN_KNOB_UNSCALED_WALK = 5,     /* 0 = the forward walk always applies the preconditioner itself (k_down_chain / k_dual_stage pair) */
    RN_KNOB_STREAM_TWO_PER_CU = 6, /* k_stream_gemv: 1 = the instantiation that leaves room for two workgroups per CU, 0 = the other */
    RN_KNOB_STREAM_SPLIT = 7,      /* k_stream_gemv: 0 = the last partial round is not split by columns */
    RN_KNOB_NAMA_PAIR = 8,         /* NAMA: 0 = its two Hessian sweeps one after the other instead of one pass over the operator blocks */
    RN_KNOB_LS_SEQUENTIAL = 9,     /* global FBE / NAMA: 1 = trial-by-trial line search instead of the batched candidates */
    RN_KNOB_VALUE_MFMA = 10,       /* global FBE / NAMA: 0 = the value's primal terms on the vector ALUs */
    RN_KNOB_TUNE_BIAS_US = 11,     /* rn_exchange_autotune, test of the selection: microseconds per iteration added to THIS rank's measured one-shot time */
    RN_KNOB_STRUCT_LINEAR = 12,    /* structured mode: 0 = the first product k_gemm_prep_m2 in front of the chain walks instead of the linear form (k_up_chain_lin); 2 = the linear form without the chain walk riding in the fused walk + dual update; 3 = the linear form with the subtree sums of beta walked and multiplied in every iteration instead of once per control step; 4 = that constant, but v_i and [L v_i; B L v_i] as two products instead of one with the composite operator; 5 = the composite operator without the forward walk's affine terms in its constant operand */
    RN_KNOB_FUSE_SPLIT = 13,       /* fused forward walk + dual update: workgroups per chain (1 .. chain length); 0 = one per chain and only where the chains fill 3/4 of the CUs (round 6's first rule) */
    RN_KNOB_COUNT = 14
};
int rn_debug_set_knob(rn_ctx *ctx, int knob, int value);

#ifdef __cplusplus
}
#endif
#endif /* RAPIDNET_DEBUG_H_ */
