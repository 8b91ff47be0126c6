// test_host_plugin.cpp -- the C++ plug-in surface (ogl_amd/host/OGLAdapter.H over tests/cpp/MiniFoam.H).
//   ./test_host_plugin cpu   host logic, dictionary / selection-table behaviour, loud failure w/o GPU
//   ./test_host_plugin gpu   GKOCG through lduMatrix::solver::New on the MI355X vs the oracle
// The first three cases re-state the reference's gtest cases (unitTests/test_HostMatrix.C:8-107)
// against this build's Foam:: free functions; the vectors are the reference's known answers.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "MiniFoam.H"
#include "OGLAdapter.H"
extern "C" {
#include "ogl_oracle.h"  // checker only
}

OGL_REGISTER_SOLVERS

using namespace Foam;

static int g_failed = 0, g_run = 0;
#define EXPECT_TRUE(c)                                                         \
    do {                                                                       \
        if (!(c)) {                                                            \
            std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);        \
            ++g_failed;                                                        \
        }                                                                      \
    } while (0)
#define EXPECT_EQ(a, b) EXPECT_TRUE((a) == (b))
#define TEST(name)                                     \
    static void name();                                \
    struct name##_reg {                                \
        name##_reg() { tests().push_back({#name, name}); } \
    } name##_inst;                                     \
    static void name()
struct TestCase {
    const char *name;
    void (*fn)();
};
static std::vector<TestCase> &tests()
{
    static std::vector<TestCase> t;
    return t;
}
template <class F>
static std::string fatal_message(F f)
{
    try {
        f();
    } catch (const Foam::error &e) {
        return e.what();
    }
    return "";
}

// ------------------------------------------------------------------ HostMatrixConversion (cpu)
TEST(cpu_HostMatrixConversion_symmetric_update)
{
    std::vector<scalar> d{1., 2., 3., 4., 5.};
    std::vector<scalar> u{10., 11., 20., 12., 21., 13.};
    std::vector<label> p{6, 0, 2, 0, 7, 1, 4, 1, 8, 3, 2, 3, 9, 5, 4, 5, 10};
    std::vector<scalar> res(17, 0.);
    std::vector<scalar> exp{1., 10., 20., 10., 2., 11., 21., 11., 3., 12., 20., 12., 4., 13., 21., 13., 5.};
    Foam::symmetric_update(17, 6, p.data(), 1.0, d.data(), u.data(), res.data());
    EXPECT_EQ(res, exp);
    Foam::symmetric_update(17, 6, p.data(), -1.0, d.data(), u.data(), res.data());
    EXPECT_EQ(res, exp);  // the reference ignores scale here (HostMatrixFreeFunctions.C:27-28)
}

TEST(cpu_HostMatrixConversion_non_symmetric_update)
{
    std::vector<scalar> d{1., 1., 1., 1., 1.};
    std::vector<scalar> u{1., 2., 1., 2., 1., 1.};
    std::vector<scalar> l{2., 2., 3., 2., 3., 2.};
    std::vector<label> p{12, 0, 1, 6, 13, 2, 3, 7, 14, 4, 8, 9, 15, 5, 10, 11, 16};
    std::vector<scalar> res(17, 0.);
    std::vector<scalar> exp{1., 1., 2., 2., 1., 1., 2., 2., 1., 1., 3., 2., 1., 1., 3., 2., 1.};
    Foam::non_symmetric_update(17, 6, p.data(), 1.0, d.data(), u.data(), l.data(), res.data());
    EXPECT_EQ(res, exp);
}

TEST(cpu_HostMatrixConversion_init_local_sparsity)
{
    std::vector<label> upper{1, 3, 2, 4, 3, 4};
    std::vector<label> lower{0, 0, 1, 1, 2, 3};
    std::vector<label> rows(17, 0), cols(17, 0), permute(17, 0);
    std::vector<label> rows_exp{0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4};
    std::vector<label> cols_exp{0, 1, 3, 0, 1, 2, 4, 1, 2, 3, 0, 2, 3, 4, 1, 3, 4};
    std::vector<label> permute_exp{6, 0, 1, 0, 7, 2, 3, 2, 8, 4, 1, 4, 9, 5, 3, 5, 10};
    Foam::init_local_sparsity(5, 6, true, upper.data(), lower.data(), rows.data(), cols.data(),
                              permute.data());
    EXPECT_EQ(rows, rows_exp);
    EXPECT_EQ(cols, cols_exp);
    EXPECT_EQ(permute, permute_exp);
}

// ------------------------------------------------------------------ a small Poisson lduMatrix
struct Case {
    objectRegistry db;
    lduMesh mesh{db};
    std::unique_ptr<lduAddressing> addr;
    std::unique_ptr<lduMatrix> A;
    FieldField<Field, scalar> bou, intc;
    std::vector<std::unique_ptr<lduInterface>> patches;
    std::vector<std::unique_ptr<lduInterfaceField>> fields;
    lduInterfaceFieldPtrsList ifaces;
    label n = 0;
};

static void build_poisson(Case &c, int nx, int ny, int nz, bool periodic_x, bool asym)
{
    std::vector<label> lo, up;
    const label N = nx * ny * nz;
    std::vector<scalar> diag(N);
    for (label cell = 0; cell < N; ++cell) {
        const int i = cell % nx, j = (cell / nx) % ny, k = cell / (nx * ny);
        int nb = (i > 0) + (i < nx - 1) + (j > 0) + (j < ny - 1) + (k > 0) + (k < nz - 1);
        if (periodic_x) nb += (i == 0) + (i == nx - 1);
        diag[cell] = nb + 1e-3 * (1.0 + (cell % 7) / 7.0);
        if (i < nx - 1) { lo.push_back(cell); up.push_back(cell + 1); }
        if (j < ny - 1) { lo.push_back(cell); up.push_back(cell + nx); }
        if (k < nz - 1) { lo.push_back(cell); up.push_back(cell + nx * ny); }
    }
    c.n = N;
    c.addr = std::make_unique<lduAddressing>(labelList(lo), labelList(up));
    const label F = static_cast<label>(lo.size());
    if (asym)
        c.A = std::make_unique<lduMatrix>(c.mesh, *c.addr, scalarField(diag), scalarField(F, -0.9),
                                          scalarField(F, -1.1));
    else
        c.A = std::make_unique<lduMatrix>(c.mesh, *c.addr, scalarField(diag), scalarField(F, -1.0));
    if (periodic_x) {
        std::vector<label> left, right;
        for (label cell = 0; cell < N; ++cell) {
            if (cell % nx == 0) left.push_back(cell);
            if (cell % nx == nx - 1) right.push_back(cell);
        }
        c.patches.push_back(std::make_unique<cyclicFvPatch>(labelList(left), 1));
        c.patches.push_back(std::make_unique<cyclicFvPatch>(labelList(right), 0));
        for (int p = 0; p < 2; ++p) {
            c.addr->setPatchAddr(p, &c.patches[p]->faceCells());
            c.fields.push_back(std::make_unique<lduInterfaceField>(*c.patches[p]));
            c.ifaces.append(c.fields.back().get());
            c.bou.push_back(scalarField(static_cast<label>(left.size()), asym && p == 1 ? 1.1 : (asym ? 0.9 : 1.0)));
            c.intc.push_back(scalarField(static_cast<label>(left.size()), 0.0));
        }
    }
}

static dictionary cg_dict()
{
    dictionary pc;
    pc.add("preconditioner", "BJ").add("maxBlockSize", 1);
    dictionary d;
    d.add("solver", "GKOCG").add("preconditioner", pc).add("tolerance", 1e-10).add("relTol", 0.0);
    d.add("maxIter", 300).add("export", "true").add("matrixFormat", "Csr").add("executor", "hip");
    d.add("adaptMinIter", "false");
    return d;
}

// ------------------------------------------------------------------ momentum components (cpu)
TEST(cpu_component_sibling_names)
{
    EXPECT_TRUE(componentSiblings("Ux", "Uy") && componentSiblings("Uz", "Ux") && componentSiblings("Rxx", "Rxy"));
    EXPECT_TRUE(componentSiblings("Rxy", "Ryz") && componentSiblings("U.waterx", "U.watery"));
    EXPECT_TRUE(!componentSiblings("Ux", "Ux") && !componentSiblings("Ux", "p") && !componentSiblings("x", "y"));
    EXPECT_TRUE(!componentSiblings("Ux", "Vx") && !componentSiblings("Ux", "Uxx") && !componentSiblings("k", "p"));
    EXPECT_TRUE(!componentSiblings("Ua", "Ub") && !componentSiblings("", ""));
    // the donor is an EARLIER component of the same solveSegregated loop: Ux never takes from the Uz before it
    EXPECT_TRUE(earlierComponent("Ux", "Uy") && earlierComponent("Uy", "Uz") && earlierComponent("Ux", "Uz"));
    EXPECT_TRUE(earlierComponent("Rxx", "Rxy") && earlierComponent("Ryz", "Rzz"));
    EXPECT_TRUE(!earlierComponent("Uz", "Ux") && !earlierComponent("Uy", "Ux") && !earlierComponent("Ux", "Ux"));
    EXPECT_TRUE(!earlierComponent("Ux", "p") && !earlierComponent("Rzz", "Rxx"));
}

// ------------------------------------------------------------------ dictionary / selection (cpu)
TEST(cpu_config_defaults_and_keywords)
{
    dictionary d;
    d.add("solver", "GKOCG").add("preconditioner", "none");
    ogl_config c = read_ogl_config(d, "GKOCG");
    EXPECT_EQ(c.solver, OGL_SOLVER_CG);
    EXPECT_EQ(c.preconditioner, OGL_PRECOND_NONE);
    EXPECT_TRUE(c.tolerance == 1e-6 && c.rel_tol == 1e-6 && c.max_iter == 1000 && c.min_iter == 0);
    EXPECT_TRUE(c.matrix_format == OGL_FORMAT_COO && c.update_rhs == 1 && c.update_init_guess == 0);
    EXPECT_TRUE(c.relaxation_factor == 0.6 && c.adapt_min_iter == 1 && c.eval_frequency == 1);
    ogl_config c2 = read_ogl_config(cg_dict(), "GKOCG");
    EXPECT_TRUE(c2.preconditioner == OGL_PRECOND_BJ && c2.max_block_size == 1 && c2.export_res == 1);
    EXPECT_TRUE(c2.matrix_format == OGL_FORMAT_CSR && c2.tolerance == 1e-10 && c2.rel_tol == 0.0);
    d.add("preconditioner", "DIC");
    EXPECT_TRUE(fatal_message([&] { read_ogl_config(d, "GKOCG"); })
                    .find("OGL does not support the preconditioner: DIC") != std::string::npos);
    dictionary e = cg_dict();
    e.add("matrixFormat", "Hybrid");
    EXPECT_TRUE(fatal_message([&] { read_ogl_config(e, "GKOCG"); }).find("Matrix format Hybrid not supported") !=
                std::string::npos);
    dictionary f;
    f.add("solver", "GKOCG");
    EXPECT_TRUE(fatal_message([&] { read_ogl_config(f, "GKOCG"); }).find("preconditioner") != std::string::npos);
    // this build's own keywords and their defaults
    EXPECT_TRUE(c.compress_indices == 1 && c.renumber == 2 && c.symmetric_half == 1 && c.krylov_dim == 0);
    dictionary g = cg_dict();
    g.add("compressIndices", "force").add("renumber", "off").add("symmetricStorage", "false").add("krylovDim", 30);
    ogl_config c3 = read_ogl_config(g, "GKOCG");
    EXPECT_TRUE(c3.compress_indices == 2 && c3.renumber == 0 && c3.symmetric_half == 0 && c3.krylov_dim == 30);
    g.add("compressIndices", "sometimes");
    EXPECT_TRUE(fatal_message([&] { read_ogl_config(g, "GKOCG"); }).find("compressIndices") != std::string::npos);
}

TEST(cpu_runtime_selection_tables)
{
    // same tables as the reference: GKOCG symmetric only; BiCGStab / GMRES both
    auto &sym = lduMatrix::solver::symMatrixConstructorTable();
    auto &asym = lduMatrix::solver::asymMatrixConstructorTable();
    EXPECT_TRUE(sym.count("GKOCG") && sym.count("GKOBiCGStab") && sym.count("GKOGMRES"));
    EXPECT_TRUE(!asym.count("GKOCG") && asym.count("GKOBiCGStab") && asym.count("GKOGMRES"));
    Case c;
    build_poisson(c, 3, 3, 3, false, true);
    dictionary d = cg_dict();
    const std::string msg = fatal_message(
        [&] { lduMatrix::solver::New("U", *c.A, c.bou, c.intc, c.ifaces, d); });
    EXPECT_TRUE(msg.find("Unknown asymmetric matrix solver GKOCG") != std::string::npos);
    d.add("executor", "reference");
    Case s;
    build_poisson(s, 3, 3, 3, false, false);
    EXPECT_TRUE(fatal_message([&] { lduMatrix::solver::New("p", *s.A, s.bou, s.intc, s.ifaces, d); })
                    .find("executor reference is not available") != std::string::npos);
}

TEST(cpu_no_device_fails_loudly)
{
    int n_dev = 0;
    ogl_registry *probe = nullptr;
    if (ogl_registry_create(&probe, 0, nullptr) == OGL_OK) {
        ogl_registry_destroy(probe);
        std::printf("  (a GPU is visible: skipped)\n");
        return;
    }
    (void)n_dev;
    Case c;
    build_poisson(c, 3, 3, 3, false, false);
    const std::string msg = fatal_message(
        [&] { lduMatrix::solver::New("p", *c.A, c.bou, c.intc, c.ifaces, cg_dict()); });
    EXPECT_TRUE(msg.find("cannot create device context") != std::string::npos);
    EXPECT_TRUE(msg.find("no CPU path") != std::string::npos);
}

// ------------------------------------------------------------------ GKOCG on the MI355X (gpu)
struct OracleSystem {
    std::vector<orc_label> rows, cols, perm, rowptr;
    std::vector<orc_scalar> vals, inv;
    std::vector<orc_iface> ifs;
    orc_dist_matrix A{};
};
static void oracle_system(const Case &c, OracleSystem &o)
{
    const lduMatrix &m = *c.A;
    const label N = c.n, F = m.lduAddr().upperAddr().size();
    for (label i = 0; i < c.ifaces.size(); ++i) {
        const auto &p = dynamic_cast<const cyclicFvPatch &>(c.ifaces[i].interface());
        o.ifs.push_back(orc_iface{ORC_IFACE_CYCLIC, -1, p.neighbPatchID(), p.faceCells().size(),
                                  p.faceCells().cdata(), c.bou[i].cdata()});
    }
    const label ifn = orc_count_interface_nnz(o.ifs.data(), (orc_label)o.ifs.size(), 0);
    const label nnz = N + 2 * F + ifn;
    o.rows.resize(nnz), o.cols.resize(nnz), o.perm.resize(nnz), o.vals.resize(nnz);
    orc_init_local_sparsity_pattern(N, F, m.symmetric(), m.lduAddr().upperAddr().cdata(),
                                    m.lduAddr().lowerAddr().cdata(), o.ifs.data(), (orc_label)o.ifs.size(),
                                    o.rows.data(), o.cols.data(), o.perm.data());
    orc_update_local_matrix_data(N, F, m.symmetric(), m.diag().cdata(), m.upper().cdata(),
                                 m.lower().cdata(), o.ifs.data(), (orc_label)o.ifs.size(), o.perm.data(),
                                 nnz, o.vals.data());
    o.rowptr.resize(N + 1);
    orc_rowptr_from_rows(N, nnz, o.rows.data(), o.rowptr.data());
    o.inv.resize(N);
    orc_jacobi_generate_scalar(N, o.rowptr.data(), o.cols.data(), o.vals.data(), o.inv.data());
    o.A.n = N;
    o.A.rowptr = o.rowptr.data();
    o.A.cols = o.cols.data();
    o.A.vals = o.vals.data();
    o.A.global_n = N;
}

static void run_gkocg_case(bool periodic)
{
    Case c;
    build_poisson(c, 12, 10, 8, periodic, false);
    scalarField source(c.n), psi(c.n, 0.0);
    for (label i = 0; i < c.n; ++i) source[i] = std::sin(0.37 * i) + 0.25;
    const dictionary d = cg_dict();
    auto solver = lduMatrix::solver::New("p", *c.A, c.bou, c.intc, c.ifaces, d);
    const solverPerformance perf = solver->solve(psi, source);
    EXPECT_EQ(perf.solverName(), word("BJhipGKOCG"));  // lduLduBase.H:315-317
    EXPECT_EQ(perf.fieldName(), word("p"));

    OracleSystem o;
    oracle_system(c, o);
    std::vector<orc_scalar> x(c.n, 0.0), hist(400, 0.0);
    orc_criterion crit{1e-10, 0.0, 0, 300, 1, 1};
    orc_criterion_state st{};
    st.history = hist.data();
    orc_set_reduction(ORC_REDUCE_BLOCKED, ogl_reduction_chunk_rows());
    orc_cg(&o.A, source.cdata(), x.data(), o.inv.data(), &crit, &st);
    EXPECT_EQ(perf.nIterations(), st.iter);
    EXPECT_EQ(perf.initialResidual(), st.init_residual);
    EXPECT_EQ(perf.finalResidual(), st.residual);
    EXPECT_TRUE(perf.finalResidual() < 1e-10);
    EXPECT_TRUE(std::memcmp(psi.cdata(), x.data(), sizeof(scalar) * c.n) == 0);
    const auto h = dynamic_cast<const GKOlduBaseSolver &>(*solver).get_res_norms();
    EXPECT_EQ((label)h.size(), st.iter);
    EXPECT_TRUE(std::memcmp(h.data(), hist.data(), sizeof(scalar) * h.size()) == 0);

    // a fresh solver object per solve finds the device state by field name; psi is NOT re-read
    // (updateInitGuess false), the new source is
    scalarField source2(c.n), psi2(c.n, 99.0);
    for (label i = 0; i < c.n; ++i) source2[i] = std::cos(0.11 * i);
    auto solver2 = lduMatrix::solver::New("p", *c.A, c.bou, c.intc, c.ifaces, d);
    const solverPerformance perf2 = solver2->solve(psi2, source2);
    std::vector<orc_scalar> x2(x);
    orc_criterion_state st2{};
    st2.history = hist.data();
    orc_cg(&o.A, source2.cdata(), x2.data(), o.inv.data(), &crit, &st2);
    orc_set_reduction(ORC_REDUCE_SEQUENTIAL, 0);
    EXPECT_EQ(perf2.nIterations(), st2.iter);
    EXPECT_TRUE(std::memcmp(psi2.cdata(), x2.data(), sizeof(scalar) * c.n) == 0);
}

TEST(gpu_GKOCG_BJ_matches_oracle) { run_gkocg_case(false); }

// the backend's switches as keywords of the solver dictionary (OGLAdapter.H)
TEST(gpu_property_keywords_reach_the_backend)
{
    Case c;
    build_poisson(c, 12, 10, 8, false, false);
    scalarField source(c.n, 1.0), psi(c.n, 0.0);
    dictionary d = cg_dict();
    auto s1 = lduMatrix::solver::New("pk_default", *c.A, c.bou, c.intc, c.ifaces, d);
    s1->solve(psi, source);
    EXPECT_EQ(dynamic_cast<const GKOlduBaseSolver &>(*s1).backend_property("fusedFinalizersInUse"), 1.0);
    d.add("fusedFinalizers", 0);
    scalarField psi2(c.n, 0.0);
    auto s2 = lduMatrix::solver::New("pk_off", *c.A, c.bou, c.intc, c.ifaces, d);
    s2->solve(psi2, source);
    EXPECT_EQ(dynamic_cast<const GKOlduBaseSolver &>(*s2).backend_property("fusedFinalizersInUse"), 0.0);
    EXPECT_TRUE(std::memcmp(psi.cdata(), psi2.cdata(), sizeof(scalar) * c.n) == 0);   // same bits either way
}
TEST(gpu_GKOCG_BJ_cyclic_patches) { run_gkocg_case(true); }

// fvMatrix<vector>::solveSegregated: ONE lduMatrix, the solvers of Ux, Uy, Uz built one after the other, diag() changed in
// between.  With `componentCoeffsReuse true` the second and third component take the first one's device copy of upper /
// lower (offDiagReused) and give the bits of a full upload; a new time step, a field that is no sibling, a LATER component
// as the donor, or off-diagonals written to in between: full upload.  Without the keyword every component uploads all
// (the reference's behaviour, HostMatrix.C:644-682).
TEST(gpu_momentum_components_share_the_off_diagonals)
{
    Case c;
    build_poisson(c, 14, 11, 9, false, true);
    dictionary pc;
    pc.add("preconditioner", "BJ").add("maxBlockSize", 1);
    dictionary d;
    d.add("solver", "GKOBiCGStab").add("preconditioner", pc).add("tolerance", 1e-10).add("relTol", 0.0);
    d.add("maxIter", 300).add("export", "true").add("matrixFormat", "Csr").add("executor", "hip").add("adaptMinIter", "false");
    dictionary d_off = d;   // (the default: no reuse)
    d.add("componentCoeffsReuse", "true");
    const scalarField diag0(c.A->diag());
    auto component = [&](const char *name, int cmpt, const dictionary &dict, scalarField &psi, double &reused) {
        for (label i = 0; i < c.n; ++i) c.A->diag()[i] = diag0[i] + 0.01 * cmpt * (1 + i % 5);   // (addBoundaryDiag)
        scalarField source(c.n);
        for (label i = 0; i < c.n; ++i) source[i] = std::sin(0.37 * i + cmpt) + 0.25;
        auto solver = lduMatrix::solver::New(name, *c.A, c.bou, c.intc, c.ifaces, dict);
        const solverPerformance perf = solver->solve(psi, source, (direction)cmpt);
        reused = dynamic_cast<const GKOlduBaseSolver &>(*solver).backend_property("offDiagReused");
        return perf;
    };
    Foam::Time::index() = 7;
    double r[3], q[3];
    std::vector<scalarField> x(3, scalarField(c.n, 0.0)), y(3, scalarField(c.n, 0.0));
    const char *names[3] = {"Ux", "Uy", "Uz"}, *names_off[3] = {"Vx", "Vy", "Vz"};
    solverPerformance px[3], py[3];
    for (int k = 0; k < 3; ++k) px[k] = component(names[k], k, d, x[k], r[k]);
    EXPECT_TRUE(r[0] == 0.0 && r[1] == 1.0 && r[2] == 1.0);
    for (int k = 0; k < 3; ++k) py[k] = component(names_off[k], k, d_off, y[k], q[k]);       // every component uploads all
    EXPECT_TRUE(q[0] == 0.0 && q[1] == 0.0 && q[2] == 0.0);
    for (int k = 0; k < 3; ++k) {
        EXPECT_EQ(px[k].nIterations(), py[k].nIterations());
        EXPECT_EQ(px[k].finalResidual(), py[k].finalResidual());
        EXPECT_TRUE(std::memcmp(x[k].cdata(), y[k].cdata(), sizeof(scalar) * c.n) == 0);
    }
    // the next time step: Ux uploads again (the previous object was Vz: no sibling), then a time-index change between
    // two siblings, then off-diagonals written to in between (the sampled checksum sees entry 0 change)
    Foam::Time::index() = 8;
    double rr;
    scalarField z(c.n, 0.0);
    component("Ux", 0, d, z, rr);
    EXPECT_EQ(rr, 0.0);
    Foam::Time::index() = 9;
    component("Uy", 1, d, z, rr);
    EXPECT_EQ(rr, 0.0);
    component("Uz", 2, d, z, rr);
    EXPECT_EQ(rr, 1.0);
    component("Ux", 0, d, z, rr);   // the next outer corrector, same time index: Uz is not an earlier component of Ux
    EXPECT_EQ(rr, 0.0);
    c.A->upper()[0] *= 1.5;
    component("Uy", 1, d, z, rr);
    EXPECT_EQ(rr, 0.0);
    component("p", 0, d, z, rr);   // (no sibling of Uy)
    EXPECT_EQ(rr, 0.0);
}

// The default plug-in path trusts nothing: ONE coefficient that no sampled checksum would look at (24,648 faces: the
// sample of ogl_solver_set_matrix_like takes every 6th entry) changes between the constructors of Ux and Uy, and Uy's
// device matrix has it (HostMatrix.C:644-682: the reference uploads every time).
TEST(gpu_an_unsampled_coefficient_change_between_components_reaches_the_device)
{
    Case c;
    build_poisson(c, 24, 20, 18, false, true);
    dictionary pc;
    pc.add("preconditioner", "BJ").add("maxBlockSize", 1);
    dictionary d;
    d.add("solver", "GKOBiCGStab").add("preconditioner", pc).add("tolerance", 1e-10).add("relTol", 0.0);
    d.add("maxIter", 300).add("matrixFormat", "Csr").add("executor", "hip").add("adaptMinIter", "false");
    Foam::Time::index() = 21;
    scalarField source(c.n);
    for (label i = 0; i < c.n; ++i) source[i] = std::sin(0.37 * i) + 0.25;
    auto solve_as = [&](const char *name, scalarField &psi, double &reused) {
        auto solver = lduMatrix::solver::New(name, *c.A, c.bou, c.intc, c.ifaces, d);
        solver->solve(psi, source, 0);
        reused = dynamic_cast<const GKOlduBaseSolver &>(*solver).backend_property("offDiagReused");
    };
    double r0, r1, r2;
    scalarField ux(c.n, 0.0), uy(c.n, 0.0), fresh(c.n, 0.0);
    solve_as("Ux", ux, r0);
    const label face = 1;                      // 1 % 6 != 0 and not the last face: outside the sample
    EXPECT_TRUE(c.A->upper().size() > 2 * 4096);
    c.A->upper()[face] *= 3.0;
    solve_as("Uy", uy, r1);
    solve_as("fresh", fresh, r2);              // a field that never had a sibling: what a full upload gives
    EXPECT_TRUE(r0 == 0.0 && r1 == 0.0 && r2 == 0.0);
    EXPECT_TRUE(std::memcmp(uy.cdata(), fresh.cdata(), sizeof(scalar) * c.n) == 0);
    EXPECT_TRUE(std::memcmp(uy.cdata(), ux.cdata(), sizeof(scalar) * c.n) != 0);
}

TEST(gpu_unsupported_coupled_patch_is_fatal)
{
    Case c;
    build_poisson(c, 4, 4, 4, false, false);
    c.patches.push_back(std::make_unique<cyclicAMIFvPatch>(labelList({0, 1})));
    c.fields.push_back(std::make_unique<lduInterfaceField>(*c.patches.back()));
    c.ifaces.append(c.fields.back().get());
    c.bou.push_back(scalarField(2, 1.0));
    c.intc.push_back(scalarField(2, 0.0));
    const std::string msg = fatal_message(
        [&] { lduMatrix::solver::New("q", *c.A, c.bou, c.intc, c.ifaces, cg_dict()); });
    EXPECT_TRUE(msg.find("unsupported coupled patch") != std::string::npos);
}

int main(int argc, char **argv)
{
    const std::string which = argc > 1 ? argv[1] : "cpu";
    for (const auto &t : tests()) {
        if (std::strncmp(t.name, which.c_str(), which.size()) != 0) continue;
        std::printf("[ RUN  ] %s\n", t.name);
        const int before = g_failed;
        try {
            t.fn();
        } catch (const std::exception &e) {
            std::printf("  EXCEPTION %s\n", e.what());
            ++g_failed;
        }
        ++g_run;
        std::printf("[ %s ] %s\n", g_failed == before ? " OK " : "FAIL", t.name);
    }
    std::printf("%d test(s) run, %d failure(s)\n", g_run, g_failed);
    return g_failed ? 1 : (g_run ? 0 : 2);
}
