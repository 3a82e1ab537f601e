// Run-time RCCL binding used by the engine's fm_comm_* / fm_fedavg_* entry points (comm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

const char* fmcomm_error();
bool fmcomm_preflight();      // dlopen librccl and resolve its entry points: local, not a collective
bool fmcomm_unique_id(unsigned char id[128]);
bool fmcomm_init(void** comm, const unsigned char id[128], int rank, int world);
bool fmcomm_destroy(void* comm);
// in-place sum over the communicator's ranks, enqueued on stream s (fp32 or fp64 elements)
bool fmcomm_allreduce_sum(void* comm, void* buf, size_t n, bool f64, hipStream_t s);
