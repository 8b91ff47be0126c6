// GKOSolvers.C -- the one translation unit an OpenFOAM tree compiles to get libOGL.so backed by
// libogl_amd.so.  NOT built in this repository's image (no OpenFOAM here); it is the adapter of
// ogl_amd/host/OGLAdapter.H compiled against the real headers instead of MiniFoam.H.
//
//   wmake libso ogl_amd/foam          (needs FOAM_SRC; see Make/options)
//   controlDict:  libs ("libOGL.so");           (README.md:63-65 of the reference)
//   fvSolution:   p { solver GKOCG; preconditioner BJ; executor hip; ... }
//
// Replaces: Solver/CG/GKOCG.C:14-17, Solver/BiCGStab/GKOBiCGStab.C:14-20,
//           Solver/GMRES/GKOGMRES.C:14-20 (registration) and everything they pull in.
#include "fvCFD.H"
#include "lduMatrix.H"
#include "processorFvPatch.H"
#include "processorLduInterface.H"
#include "cyclicFvPatch.H"
#include "cyclicAMIFvPatch.H"
#include "regIOobject.H"
#include "Pstream.H"
#include "PstreamReduceOps.H"

// OpenFOAM spells these differently from the stand-in:
#define OGL_ABORT_FATAL Foam::abort(Foam::FatalError)
// `debug true`: export at write times into processor?/<time>/ (lduLduBase.H:259-264, common.C:31-45)
#define OGL_EXPORT_NOW(db) ((db).time().writeTime())
#define OGL_EXPORT_DIR(db) (Foam::mkDir((db).time().timePath()), std::string((db).time().timePath()))

// mesh.C() for ogl_ldu_view::cell_centres (three contiguous doubles per cell): with `renumber auto | on` the cells along
// a Hilbert curve through their centres compete with reverse Cuthill-McKee for the numbering of the device copy
#define OGL_CELL_CENTRES(m)                                                                                     \
    (Foam::isA<Foam::fvMesh>((m).mesh())                                                                        \
         ? reinterpret_cast<const double *>(Foam::refCast<const Foam::fvMesh>((m).mesh()).C().primitiveField().cdata()) \
         : nullptr)

#include "OGLAdapter.H"

namespace Foam {

// Communicator for decomposed runs (ExecutorHandler.H:140-144,167-172).
//  * default: RCCL over xGMI -- rank 0 creates the unique id, Pstream broadcasts the 128 bytes;
//  * forceHostBuffer true: the library stages halo/scalars through pinned host memory and these
//    callbacks move them with OpenFOAM's own message passing.
namespace {

void ogl_allreduce_sum(void *, double *v, int32_t n)
{
    for (int32_t i = 0; i < n; ++i) reduce(v[i], sumOp<scalar>());
}

void ogl_neighbour_exchange(void *, int32_t n_nbr, const int32_t *ranks, const int32_t *counts,
                            const double *send, double *recv)
{
    label off = 0;
    PstreamBuffers bufs(Pstream::commsTypes::nonBlocking);
    for (int32_t i = 0; i < n_nbr; ++i) {
        UOPstream os(ranks[i], bufs);
        os.write(reinterpret_cast<const char *>(send + off), counts[i] * sizeof(double));
        off += counts[i];
    }
    bufs.finishedSends();
    off = 0;
    for (int32_t i = 0; i < n_nbr; ++i) {
        UIPstream is(ranks[i], bufs);
        is.read(reinterpret_cast<char *>(recv + off), counts[i] * sizeof(double));
        off += counts[i];
    }
}

struct InstallCommHook {
    InstallCommHook()
    {
        OGLDeviceRegistry::commHook() = [](ogl_registry *reg, const dictionary &controls) {
            const bool host = controls.lookupOrDefault<Switch>("forceHostBuffer", false);
            int rc = OGL_OK;
            label rccl_ok = 0;
            if (!host) {
                // RCCL over xGMI: init runs a collective self-test (one all-reduce, one ring of send / recv with
                // known values).  A transport that fails it is not used: all ranks step down together to the
                // host-buffer transport, in this process (nothing is re-exec'ed once the GPU was touched).
                List<char> id(OGL_RCCL_ID_BYTES, '\0');
                if (Pstream::master() && ogl_rccl_unique_id(id.begin()) != OGL_OK)
                    FatalErrorInFunction << ogl_last_error() << abort(FatalError);
                Pstream::scatter(id);
                // init_rccl is collective and has no time-out: agree FIRST that every rank can enter it (library
                // loaded, device selected) -- a rank that cannot would leave the others inside ncclCommInitRank
                rccl_ok = ogl_registry_rccl_ready(reg) == OGL_OK;
                if (!rccl_ok)
                    WarningInFunction << "RCCL transport unavailable on this rank: " << ogl_last_error() << endl;
                reduce(rccl_ok, minOp<label>());
                if (rccl_ok) {
                    // past this point a local failure means the collective itself broke on this rank while others
                    // may still be inside it: abort the job, as the reference does (uncaught Ginkgo exception)
                    const int rci =
                        ogl_registry_init_rccl(reg, Pstream::myProcNo(), Pstream::nProcs(), id.begin());
                    if (rci == OGL_ERR_COMM_SELFTEST) {
                        rccl_ok = 0;  // the collectives completed and delivered wrong numbers: every rank sees that
                        WarningInFunction << "RCCL self-test failed: " << ogl_last_error() << endl;
                    } else if (rci != OGL_OK) {
                        FatalErrorInFunction << ogl_last_error() << abort(FatalError);
                    }
                    reduce(rccl_ok, minOp<label>());
                }
                if (!rccl_ok) WarningInFunction << "using the host-buffer transport (forceHostBuffer)" << endl;
            }
            if (!rccl_ok)
                rc = ogl_registry_set_host_comm(reg, Pstream::myProcNo(), Pstream::nProcs(),
                                                ogl_allreduce_sum, ogl_neighbour_exchange, nullptr);
            if (rc != OGL_OK) FatalErrorInFunction << ogl_last_error() << abort(FatalError);
            // Scalar all-reduces and halo puts through peer-written memory over xGMI (hipIpc), on top of the
            // transport above (which stays the bootstrap and the fallback): OPT-IN (`peerAllReduce true`) until a
            // run with one rank per device is on record -- every multi-rank run so far shared one GPU, and the
            // connect-time self-test does not exercise the wait inside the SpMV kernel under real xGMI latency.
            // Connecting runs a collective self-test (two all-reduces through the mailboxes,
            // one put / wait round over the ring) and any rank that cannot join makes all ranks keep the
            // transport's own all-reduce and halo exchange.  Only tried when every rank sits on the same host
            // (hipIpc does not cross nodes; without this check the ranks of a multi-node run would sit in the 60 s
            // self-test time-out before falling back) and with at most 16 ranks.
            bool one_host = true;
            {
                List<word> hosts(Pstream::nProcs());
                hosts[Pstream::myProcNo()] = hostName();
                Pstream::gatherList(hosts);
                Pstream::scatterList(hosts);
                forAll(hosts, p) one_host = one_host && hosts[p] == hosts[0];
            }
            if (controls.lookupOrDefault<Switch>("peerAllReduce", false) && one_host &&
                Pstream::nProcs() <= 16) {
                List<List<char>> handles(Pstream::nProcs());
                handles[Pstream::myProcNo()].setSize(OGL_PEER_HANDLE_BYTES, '\0');
                label ok = ogl_registry_peer_handle(reg, handles[Pstream::myProcNo()].begin()) == OGL_OK;
                Pstream::gatherList(handles);
                Pstream::scatterList(handles);
                reduce(ok, minOp<label>());
                if (ok) {
                    List<char> flat(OGL_PEER_HANDLE_BYTES * Pstream::nProcs());
                    forAll(handles, p)
                        std::copy(handles[p].begin(), handles[p].end(),
                                  flat.begin() + OGL_PEER_HANDLE_BYTES * p);
                    ok = ogl_registry_peer_connect(reg, Pstream::myProcNo(), Pstream::nProcs(),
                                                   flat.begin()) == OGL_OK;
                    reduce(ok, minOp<label>());
                }
                if (!ok) ogl_registry_peer_disable(reg);
            }
        };
    }
} installCommHook_;

}  // namespace

defineTypeNameAndDebug(GKOCG, 0);
defineTypeNameAndDebug(GKOBiCGStab, 0);
defineTypeNameAndDebug(GKOGMRES, 0);

}  // namespace Foam

OGL_REGISTER_SOLVERS
