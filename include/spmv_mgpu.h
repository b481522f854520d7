/*
 * spmv_mgpu.h -- single-process multi-GPU SpMV over the GPUs of one node:
 * contiguous row ranges (boundaries multiples of HACK_SIZE), x replicated,
 * every device computes its fragment of y, fragments exchanged with one
 * grouped in-place ncclAllGather (RCCL over xGMI).  The reference has no
 * multi-GPU path; this is what `spmv_scpa_amd -g N` runs.  (bench.py does
 * the same with one process per GPU through torch.distributed.)
 *
 * Partitions: equal row counts (default), or near-equal ENTRY
 * counts -- the multi-GPU form of the reference's partition_csr_rows
 * (csr.c:218-276; partition_rows_nnz_aligned, csr.h).  With the latter the
 * devices own different row counts, y holds exactly M rows and the ragged
 * fragments travel by grouped ncclSend/ncclRecv (default), one ncclBroadcast
 * per device, or one ncclAllGather padded to the longest fragment + a
 * compaction copy (spmv_mgpu_set_ragged_exchange).
 *
 * All functions return 0 or a negative errno (-ENODEV: fewer GPUs than
 * asked for).  Every entry point restores the caller's current HIP device; a
 * handle may be loaded / generated again (the previous shards are freed).
 *
 * STATUS: UNMEASURED with more than one device -- the boxes this was built
 * on have one GPU.  What HAS run there: everything with n = 1 (`-g 1`,
 * bench.py --native-mgpu --gpus 1); with spmv_mgpu_set_exchange(chunks,
 * force = 1) the RCCL side as 1-rank collectives -- the in-place all-gather,
 * the staged pipeline (chunk kernels or logical shards into the staging
 * buffer, grouped ncclAllGather per chunk on the second stream, the copy
 * back), ncclBroadcast / padded all-gather of a ragged partition -- against
 * the oracle; and on REHEARSAL handles (spmv_mgpu_create_rehearsal: n logical
 * devices on the one card, copies instead of collectives) everything that
 * depends on n > 1 but not on RCCL: even / nnz partitions, ragged offsets,
 * empty ranges, the concurrent per-device selector, pick propagation, logical
 * shards (tests/test_gpu_mgpu.py).  The n > 1 timings and RCCL's behaviour
 * across devices have not run yet.
 */
#ifndef SPMV_MGPU_H
#define SPMV_MGPU_H

#include <stddef.h>
#include <stdint.h>

#include "csr.h"
#include "spmv_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spmv_mgpu spmv_mgpu;

/* devices 0..ngpus-1, one RCCL communicator (ncclCommInitAll), one stream each */
int spmv_mgpu_create(int ngpus, spmv_mgpu **out);
void spmv_mgpu_destroy(spmv_mgpu *g);
/* REHEARSAL handle, test infrastructure: `nlogical` (<= 64) logical devices
 * spread over the visible card(s) -- all on ONE card of a 1-GPU box -- and NO
 * communicator (RCCL refuses two ranks on one device): the exchange is plain
 * device-to-device copies between the logical devices' y buffers.  Partition,
 * row offsets, empty ranges, per-device launches and the assembled y run
 * exactly as on a node; the RCCL calls do not, and its timings mean nothing.
 * spmv_mgpu_comm_ranks() returns 0 for it. */
int spmv_mgpu_create_rehearsal(int nlogical, spmv_mgpu **out);

enum spmv_mgpu_partition_kind {
    SPMV_MGPU_PART_EVEN = 0, /* equal row counts (rounded up to 32) */
    SPMV_MGPU_PART_NNZ = 1   /* near-equal entry counts, 32-aligned cuts */
};
enum spmv_mgpu_ragged_exchange {
    SPMV_MGPU_XCHG_P2P = 0,   /* grouped ncclSend / ncclRecv, in place */
    SPMV_MGPU_XCHG_BCAST = 1, /* one in-place ncclBroadcast per device */
    SPMV_MGPU_XCHG_PADDED = 2 /* ncclAllGather of `longest` rows + compaction */
};

/* shard a host matrix by rows; as_hll != 0: col-major HLL shards (kernel 1/2) */
int spmv_mgpu_load_csr(spmv_mgpu *g, const sparse_csr *A, int as_hll);
/* ... with the partition named (spmv_mgpu_load_csr = _PART_EVEN).  A range
 * may be empty when the matrix has fewer 32-row blocks than devices. */
int spmv_mgpu_load_csr_part(spmv_mgpu *g, const sparse_csr *A, int as_hll,
                            int partition);
/* weak scaling: every device generates rows_per_gpu (multiple of 32) rows of
 * the (rows_per_gpu*ngpus)-square synthetic matrix (spmv_synth.h) */
int spmv_mgpu_generate(spmv_mgpu *g, int kind, int rows_per_gpu, int K,
                       int64_t W, uint64_t seed, int as_hll);
/* ... or, _PART_NNZ, its share of the ENTRIES of that matrix (cut from the
 * generator's row lengths, partition_synth_rows_nnz) */
int spmv_mgpu_generate_part(spmv_mgpu *g, int kind, int rows_per_gpu, int K,
                            int64_t W, uint64_t seed, int as_hll,
                            int partition);

/* Which ENGINE moves the fragments of y.  _RCCL (default): collectives --
 * kernels that run on compute units.  _COPY: every device pushes its fragment
 * into its peers' y with hipMemcpyAsync (the copy engines over xGMI; peer
 * access is enabled at create): nothing competes with the SpMV kernels for
 * CUs -- what the persistent sweep launch of the blocked path wants -- any
 * partition (even or ragged) needs no staging, and with logical shards the
 * pushes of shard c run under the kernel of shard c + 1.  Not the all-gather
 * BASELINE names: an alternative bench.py --native-mgpu times beside it
 * (exchange_alternatives_ms "copy").  Rehearsal handles always use it. */
enum spmv_mgpu_engine { SPMV_MGPU_ENGINE_RCCL = 0, SPMV_MGPU_ENGINE_COPY = 1 };
int spmv_mgpu_set_exchange_engine(spmv_mgpu *g, int engine);

/* LOGICAL SHARDS: from the next load / generate on, every device holds its
 * rows as `shards` matrices of rows / shards rows each (1 = off, up to 16;
 * even partition, rows per device divisible by shards * 32, else the load
 * answers -EINVAL).  A kernel that runs whole matrices only -- the blocked
 * path -- then overlaps all the same: logical shard c of every device is
 * all-gathered on the second stream while shard c + 1 computes (the staged
 * pipeline with chunk = shard).  reserve_cus: compute units a sweep copy
 * leaves OUT of its persistent grid so that RCCL's kernels run beside it
 * (spmv_mgpu_autotune rebuilds the picked sweep layout with it when
 * shards > 1).  What bench.py --native-mgpu times against the plain
 * arrangement, keeping the faster (config.exchange_arrangement). */
int spmv_mgpu_set_logical_shards(spmv_mgpu *g, int shards, int reserve_cus);

/* how RAGGED fragments travel (enum spmv_mgpu_ragged_exchange; default
 * _XCHG_P2P).  Persists over reloads; the even partition keeps its in-place
 * all-gather.  With force (spmv_mgpu_set_exchange) _BCAST and _PADDED run
 * with one device too (1-rank collectives); _P2P has no peer then. */
int spmv_mgpu_set_ragged_exchange(spmv_mgpu *g, int kind);

/* the partition in use: starts[ngpus+1] row offsets and entries[ngpus] true
 * entries per device (either may be NULL); returns 1 when the ranges differ
 * (ragged fragments), 0 for the even partition, < 0 on a bad handle */
int spmv_mgpu_partition(const spmv_mgpu *g, int *starts, int64_t *entries);

int spmv_mgpu_set_x(spmv_mgpu *g, const double *x_host); /* N doubles */
int spmv_mgpu_fill_x(spmv_mgpu *g, uint64_t seed);       /* synth_x on device */

/* warmup + iters steps of (local kernels, all-gather of y); ms_each[iters]
 * = wall time per step, all devices synchronised on both sides.
 * kernel < 0: default of the shard format. */
int spmv_mgpu_spmv(spmv_mgpu *g, int kernel, int warmup, int iters,
                   double *ms_each);

/* The bench shape: `warmup` untimed steps, then EXACTLY `steps` steps enqueued
 * back to back between two all-device synchronisations; *wall_ms_total = the
 * host clock across them (one process drives all devices, so this is the
 * job's time), kernel_ms_avg[ngpus] = each device's mean event-timed kernel
 * per step (events on the device's stream around the shard launch; may be
 * NULL).  What `bench.py --native-mgpu` times. */
int spmv_mgpu_run(spmv_mgpu *g, int kernel, int warmup, int steps,
                  double *wall_ms_total, double *kernel_ms_avg);

/* How the fragments of y travel (even partition).  chunks = 1 (default, and
 * what `spmv_scpa_amd -g N` and bench.py --native-mgpu run unless asked
 * otherwise): one grouped in-place ncclAllGather after the shard kernels.  chunks = 2..16: the shard's rows in
 * that many equal pieces; chunk c of every device is all-gathered on a second
 * stream (chunk-major staging buffer, one strided copy back at the end) while
 * the kernel of chunk c+1 runs -- direct kernels only; the blocked path runs
 * whole shards and keeps chunks = 1, as does a shard whose rows do not split
 * into chunks of whole hack blocks.  Call after load / generate (it sizes the
 * staging buffer).  force != 0 runs the collective with ONE device too (a
 * 1-rank all-gather): the staging logic can then be tested on a 1-GPU box. */
int spmv_mgpu_set_exchange(spmv_mgpu *g, int chunks, int force);

/* the all-gather of y by itself, `iters` times; *ms_avg per exchange (0 with
 * one device: there is no exchange) */
int spmv_mgpu_exchange_only(spmv_mgpu *g, int iters, double *ms_avg);

/* RCCL as linked (ncclGetVersion code, e.g. 22203), ranks of the
 * communicator (ncclCommCount of device 0's), PCI bus id of device `rank` */
int spmv_mgpu_rccl_version(void);
int spmv_mgpu_comm_ranks(const spmv_mgpu *g);
int spmv_mgpu_device_bus_id(const spmv_mgpu *g, int rank, char *buf, size_t len);

/* shard `rank`: stored slots (HLL) or entries (CSR), algorithmic bytes per
 * launch, one-line layout of its blocked copy ("" when none) */
int spmv_mgpu_shard_info(const spmv_mgpu *g, int rank, int64_t *stored,
                         int64_t *alg_bytes, char *layout, size_t len);

/* measured kernel choice for the loaded shards (spmv_*_autotune per device,
 * device 0's pick for all); pass *kernel to spmv_mgpu_spmv() */
int spmv_mgpu_autotune(spmv_mgpu *g, int *kernel);

/* the 2-D blocked copy (spmv_engine.h) on every shard with one set of build
 * options (NULL: defaults): needed before naming the blocked kernel id in
 * spmv_mgpu_spmv / _run without spmv_mgpu_autotune */
int spmv_mgpu_build_panels(spmv_mgpu *g, const spmv_panel_opts *opts);

/* the gathered y as device `rank` holds it (M doubles, row order) */
int spmv_mgpu_get_y(spmv_mgpu *g, int rank, double *y_host);

/* rows_per_gpu = the longest range; bytes_per_gpu = algorithmic bytes of the
 * heaviest shard (what bounds a step) */
int spmv_mgpu_info(const spmv_mgpu *g, int *ngpus, int *rows_per_gpu,
                   int64_t *nnz_total, int64_t *bytes_per_gpu);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_MGPU_H */
