// kvm_sharded_dev.h -- what the group layer (kvm_sharded.cpp: split, slot loops, exchange protocol, merge) needs from HIP and
// RCCL, and nothing else: events, buffers, one readback kernel, one all-gather (the streams are the contexts' own).  kvm_sharded_dev.hip implements it on the real
// runtime; tests/harness/sharded_harness.cpp implements it on host memory, so that the protocol runs under ThreadSanitizer and
// ASan on a box without a GPU.  Every function returns KVM_OK or a negative status and leaves a text in kvm_sd_last_error()
// (thread-local).  Handles are opaque (hipStream_t, hipEvent_t, ncclComm_t behind void *).
#pragma once
#include <stddef.h>
#include <stdint.h>

int kvm_sd_stream_synchronize(int device, void *stream);
int kvm_sd_stream_wait_event(int device, void *stream, void *event);   // device-side: the stream goes on once the event has completed
int kvm_sd_event_create(int device, void **event);
void kvm_sd_event_destroy(int device, void *event);
int kvm_sd_event_record(int device, void *event, void *stream);
int kvm_sd_event_synchronize(void *event);
int kvm_sd_malloc(int device, void **p, size_t bytes);
void kvm_sd_free(int device, void *p);
int kvm_sd_host_alloc(int device, void **p, size_t bytes);  // pinned, readable and writable by kernels of `device`
void kvm_sd_host_free(void *p);
int kvm_sd_copy_d2h_async(int device, void *dst_host, const void *src_device, size_t bytes, void *stream);
// `blocks` blocks of `block_doubles` fp64 values in device memory, each {count, 0, count x (offset, distance) ...}: the header
// and the pairs actually present (count clamped to what the block holds; a negative count -- the failure mark -- as it is) of
// every block into the same layout in pinned host memory.  ONE kernel writing through the mapped host pointer instead of a
// copy command: answers are rare, a block is 64 KB, and a device-to-host copy costs 30 - 40 us of stream time whatever it moves.
int kvm_sd_readback_blocks(int device, double *dst_host, const double *src_device, int blocks, size_t block_doubles, void *stream);

// RCCL (librccl.so.1, opened on first use).  One communicator per local device.
int kvm_sd_unique_id(void *id, int32_t id_bytes);
int kvm_sd_comm_init_all(const int *devices, int n, void **comms);
int kvm_sd_comm_init_rank(int device, int rank, int world, const void *id, void **comm);
void kvm_sd_comm_destroy(int device, void *comm);
// ONE all-gather over the group: local device i sends `doubles` fp64 values from send[i], recv[i] receives world * doubles;
// the n local calls are issued as one RCCL group (ncclGroupStart / ncclGroupEnd), each on its own stream.
int kvm_sd_all_gather(int n, const int *devices, void *const *comms, const void *const *send, void *const *recv, size_t doubles,
                      void *const *streams);
const char *kvm_sd_last_error();
