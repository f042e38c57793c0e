// Transport plug-in seam of the row-sharded solve (dist.h).
//
// The product's transport is RCCL.  A rank's unique id that starts with "POGS" is instead handed to the shared
// object named by the environment variable POGS_AMD_TRANSPORT_PLUGIN (without it such an id is refused): the
// object exports `pogs_amd_transport()` returning this table.  The test-suite's communicators live there
// (tests/transport/test_transport.hip: ranks as threads of one process, or as processes joined by POSIX shared
// memory, on ONE GPU -- RCCL refuses two ranks on one device) and NOT in libpogs_amd.so.
//
// Contract of allreduce: sum over ranks, every rank forms the identical sum (rank order), stream semantics of
// ncclAllReduce (the result is ready for work enqueued on `stream` afterwards; `in` may equal `out`).
#pragma once
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POGS_AMD_TRANSPORT_ABI 1

typedef struct PogsAmdTransportApi {
  int abi;                                                  /* POGS_AMD_TRANSPORT_ABI */
  /* collective: every rank calls with the same 128-byte id; NULL on failure (see last_error) */
  void *(*open)(const char *unique_id, int rank, int world);
  /* 0 on success; dtype 0 = float, 1 = double; stream is a hipStream_t */
  int (*allreduce)(void *h, int rank, const void *in, void *out, size_t count, int dtype, void *stream);
  void (*close)(void *h);
  const char *(*last_error)(void);                          /* of the calling thread */
} PogsAmdTransportApi;

typedef const PogsAmdTransportApi *(*PogsAmdTransportEntry)(void);   /* symbol: pogs_amd_transport */

#ifdef __cplusplus
}
#endif
