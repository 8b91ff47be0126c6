// comm.hpp -- inter-rank transport for the sharded path (one process per GPU).
// Replaces gko::experimental::mpi::communicator + sparse_communicator
// (ExecutorHandler.H:29-32,140-144,167-172; CsrMatrixWrapper.H:195-204):
//   * scalar all-reduces for dot / norm1 / mean          
//   * neighbourhood halo exchange per SpMV               
// Two transports:
//   RcclComm -- RCCL over xGMI, device buffers end to end (the production path)
//   HostComm -- "forceHostBuffer" (ExecutorHandler.H:136-139): device data is staged through
//               pinned host memory and the HOST application moves it (MPI in OpenFOAM)
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "common.hpp"
#include "ledger.hpp"

namespace ogl {

class Comm {
public:
    virtual ~Comm() = default;
    int rank = 0;
    int n_ranks = 1;
    bool multi() const { return n_ranks > 1; }
    // In-place SUM over ranks of n doubles at device address `dev`, ordered with stream `st`.
    virtual int allreduce(double *dev, int n, hipStream_t st) = 0;
    // send/recv: device buffers blocked by neighbour (ascending rank), counts[i] doubles each way.
    virtual int exchange(const double *send, double *recv, const std::vector<int> &neighbours,
                         const std::vector<int> &counts, hipStream_t st) = 0;
    virtual const char *name() const = 0;
    // ranks the transport itself reports (RCCL: ncclCommCount), for the bench's self-check
    virtual int ranks_seen() const { return n_ranks; }
};

// Single rank: nothing to do.  (The reference refuses to run serial, lduLduBase.H:321-329; this
// build does not.)
class SelfComm final : public Comm {
public:
    int allreduce(double *, int, hipStream_t) override { return OGL_OK; }
    int exchange(const double *, double *, const std::vector<int> &, const std::vector<int> &,
                 hipStream_t) override
    {
        return OGL_OK;
    }
    const char *name() const override { return "self"; }
};

class HostComm final : public Comm {
public:
    HostComm(int rank, int n_ranks, ogl_allreduce_sum_fn ar, ogl_neighbour_exchange_fn ex,
             void *user);
    ~HostComm() override;
    int allreduce(double *dev, int n, hipStream_t st) override;
    int exchange(const double *send, double *recv, const std::vector<int> &neighbours,
                 const std::vector<int> &counts, hipStream_t st) override;
    const char *name() const override { return "host-buffer"; }

private:
    int reserve(size_t doubles);
    ogl_allreduce_sum_fn ar_;
    ogl_neighbour_exchange_fn ex_;
    void *user_;
    double *pin_send_ = nullptr, *pin_recv_ = nullptr;
    size_t cap_ = 0;
};

class RcclComm final : public Comm {
public:
    RcclComm() = default;
    ~RcclComm() override;
    static int library_ready();  // (local: librccl loads and has every symbol)
    static int unique_id(void *id_out);
    int init(int rank, int n_ranks, const void *id);
    // COLLECTIVE, right after init: one ncclAllReduce and one ring of ncclSend / ncclRecv with known values over
    // the links the solves will use; anything but the expected numbers fails loudly (OGL_ERR_COMM) here, not as
    // a wrong residual later
    int self_test(hipStream_t st);
    int allreduce(double *dev, int n, hipStream_t st) override;
    int exchange(const double *send, double *recv, const std::vector<int> &neighbours,
                 const std::vector<int> &counts, hipStream_t st) override;
    const char *name() const override { return "rccl"; }
    int ranks_seen() const override;

private:
    void *comm_ = nullptr;  // ncclComm_t
};

}  // namespace ogl
