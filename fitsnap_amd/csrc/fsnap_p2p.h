// fsnap_p2p.h -- the peer-to-peer transport of the multi-GPU exchange step (fsnap_p2p.cpp), as fsnap_comm.cpp sees it.
#pragma once

#include <cstddef>
#include <cstdint>

struct fsnap_ctx;

namespace fsnap {

struct P2P;

int p2p_make_id(char* id);                 // FSNAP_COMM_ID_BYTES bytes: magic + random token (names the shared-memory segment)
bool p2p_is_id(const char* id);
int p2p_init(fsnap_ctx* ctx, int nranks, int rank, const char* id, P2P** out);       // collective, bounded
void p2p_destroy(fsnap_ctx* ctx, P2P* p, bool broken);
bool p2p_failed(const P2P* p);             // a kernel's bounded wait ran out (its results are NaN)
int p2p_allreduce_device(fsnap_ctx* ctx, P2P* p, double* d_buf, int64_t n);          // in place, asynchronous on the context's stream
int p2p_allgather_host(fsnap_ctx* ctx, P2P* p, const void* send, size_t nbytes, void* recv);
int p2p_bcast_host(fsnap_ctx* ctx, P2P* p, void* buf, size_t nbytes, int root);
int p2p_allreduce_host(fsnap_ctx* ctx, P2P* p, double* buf, int64_t n, int op);
int p2p_barrier(fsnap_ctx* ctx, P2P* p);

}  // namespace fsnap
