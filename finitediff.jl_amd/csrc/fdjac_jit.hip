// Runtime compilation of a caller's ROW FUNCTOR (include/fdjac.h, fd_f_compile_rows).
//
// The reference calls any Julia callable as f! (src/jacobians.jl:541,563,605-606,634).  The one-launch path of this library needs f!
// as device code: a functor  T f(long long r, const P &X)  that evaluates ONE row r of the residual at the point X (X(j) = coordinate
// j) -- inside fd_csc_store_cols (include/fdjac_device.h) it evaluates, for every stored entry of every column, the entry's row at
// the column's colour point, subtracts the row at x, divides by the step and stores into nzval (src/jacobians.jl:562-568 +
// ext/FiniteDiffSparseArraysExt.jl:38-47 in one launch).  A caller without an offline toolchain (a Julia process) hands the functor
// over as SOURCE: it is compiled here with hiprtc against the embedded include/fdjac_device.h, -ffp-contract=off as the library
// itself is built (the reference never fuses a*b+c), for the device's gfx950, cached by content for the life of the process.
//
// hiprtc is bound at run time (dlopen), like RCCL: libfdjac loads on boxes without it and fd_f_compile_rows says so.
#include <dlfcn.h>
#include <hip/hiprtc.h>

#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "fdjac_internal.h"

#ifndef FDJAC_F32   /* element-type independent (the element type is a parameter of the compilation): compiled once */

namespace fdjac {

static const char kDeviceHeader[] =
#include "fdjac_device_h.inc"
    ;

struct Hiprtc {
    void *handle = nullptr;
    decltype(&hiprtcCreateProgram) CreateProgram = nullptr;
    decltype(&hiprtcDestroyProgram) DestroyProgram = nullptr;
    decltype(&hiprtcCompileProgram) CompileProgram = nullptr;
    decltype(&hiprtcAddNameExpression) AddNameExpression = nullptr;
    decltype(&hiprtcGetLoweredName) GetLoweredName = nullptr;
    decltype(&hiprtcGetProgramLogSize) GetProgramLogSize = nullptr;
    decltype(&hiprtcGetProgramLog) GetProgramLog = nullptr;
    decltype(&hiprtcGetCodeSize) GetCodeSize = nullptr;
    decltype(&hiprtcGetCode) GetCode = nullptr;
    decltype(&hiprtcGetErrorString) GetErrorString = nullptr;
    // linking a caller's LLVM bitcode in (fd_f_link_rows_bitcode): optional -- an older hiprtc without them serves source functors only
    decltype(&hiprtcGetBitcodeSize) GetBitcodeSize = nullptr;
    decltype(&hiprtcGetBitcode) GetBitcode = nullptr;
    decltype(&hiprtcLinkCreate) LinkCreate = nullptr;
    decltype(&hiprtcLinkAddData) LinkAddData = nullptr;
    decltype(&hiprtcLinkComplete) LinkComplete = nullptr;
    decltype(&hiprtcLinkDestroy) LinkDestroy = nullptr;
};
static Hiprtc g_rtc;
static std::mutex g_jit_mutex;
static thread_local std::string t_log;

static const Hiprtc *hiprtc()
{
    if (g_rtc.handle) return &g_rtc;
    const char *env = getenv("FDJAC_HIPRTC_LIB");
    const char *names[] = {env && *env ? env : "libhiprtc.so", "libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
    void *h = nullptr;
    for (const char *n : names)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        set_error("hiprtc not found (tried libhiprtc.so, /opt/rocm/lib; set FDJAC_HIPRTC_LIB): %s", dlerror());
        return nullptr;
    }
    Hiprtc r;
    r.handle = h;
#define FD_SYM(field, name)                                            \
    r.field = (decltype(r.field))dlsym(h, name);                       \
    if (!r.field) {                                                    \
        set_error("hiprtc symbol %s missing", name);                   \
        return nullptr;                                                \
    }
    FD_SYM(CreateProgram, "hiprtcCreateProgram")
    FD_SYM(DestroyProgram, "hiprtcDestroyProgram")
    FD_SYM(CompileProgram, "hiprtcCompileProgram")
    FD_SYM(AddNameExpression, "hiprtcAddNameExpression")
    FD_SYM(GetLoweredName, "hiprtcGetLoweredName")
    FD_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize")
    FD_SYM(GetProgramLog, "hiprtcGetProgramLog")
    FD_SYM(GetCodeSize, "hiprtcGetCodeSize")
    FD_SYM(GetCode, "hiprtcGetCode")
    FD_SYM(GetErrorString, "hiprtcGetErrorString")
#undef FD_SYM
    r.GetBitcodeSize = (decltype(r.GetBitcodeSize))dlsym(h, "hiprtcGetBitcodeSize");
    r.GetBitcode = (decltype(r.GetBitcode))dlsym(h, "hiprtcGetBitcode");
    r.LinkCreate = (decltype(r.LinkCreate))dlsym(h, "hiprtcLinkCreate");
    r.LinkAddData = (decltype(r.LinkAddData))dlsym(h, "hiprtcLinkAddData");
    r.LinkComplete = (decltype(r.LinkComplete))dlsym(h, "hiprtcLinkComplete");
    r.LinkDestroy = (decltype(r.LinkDestroy))dlsym(h, "hiprtcLinkDestroy");
    g_rtc = r;
    return &g_rtc;
}

// one compiled translation unit: the plain row launcher and the storing kernels of one functor type / element type
struct JitModule {
    hipModule_t mod = nullptr;
    hipFunction_t rows = nullptr;
    hipFunction_t store[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};       // [colour bytes == 4][central]
    hipFunction_t store_win[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    hipFunction_t store_ents[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};     // fd_csc_store_ents, same indices (a thread per entry)
    hipFunction_t store_rows[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};     // fd_csc_store_rows (separable functors, fd_f_compile_terms): [colour bytes == 4][central]
    unsigned lists_offset = 0, terms_bytes = 0;       // fd_sep_rows<TF>: where its two list pointers sit, sizeof(TF)
    hipFunction_t band[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};      // fd_band_store_cols: [bandwidths (1,1) / (2,2)][forward / central]
    // other bandwidths: compiled when a plan first hands this functor such a band (fd_band_store_cols<.., L, U>), kept with the module
    struct BandExtra {
        hipModule_t mod = nullptr;
        hipFunction_t fn[2] = {nullptr, nullptr};      // forward / central
        bool failed = false;
    };
    std::map<std::pair<int, int>, BandExtra> extra;
    // the complex step (src/jacobians.jl:623-648): the functor instantiated on fd_cplx<T> -- compiled when first asked for (a functor that
    // names the element type explicitly is fine for forward / central differences and fails HERE, with the compiler's message)
    struct Cplx {
        hipModule_t mod = nullptr;
        hipFunction_t rows = nullptr;                  // fdjit_rows_cplx: the plain launcher on materialised complex points
        hipFunction_t store[2] = {nullptr, nullptr};   // fd_csc_store_cols_cplx: [colour bytes == 4]
        hipFunction_t colrange[2] = {nullptr, nullptr};   // fd_colrange_store_cols<.., 2, ..>: [colour bytes == 4] (BlockBandedMatrix data)
        bool tried = false, ok = false;
        std::string log;
    } cplx;
    // column-range storage (BlockBandedMatrix data) for forward / central differences: fd_colrange_store_cols, compiled on first use
    struct ColRange {
        hipModule_t mod = nullptr;
        hipFunction_t fn[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};      // [colour bytes == 4][central]
        hipFunction_t bbb[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};     // fd_bbb_store_cols (BandedBlockBandedMatrix data), same indices
        bool tried = false, ok = false;
    } cr;
    std::vector<char> bitcode;                         // fd_f_link_rows_bitcode: the caller's LLVM bitcode, linked into every program of this functor
    std::string real;                                  // "double" / "float"
    unsigned sizeof_f = 0;
    int refs = 0;
    std::string key;              // device ordinal + '\n' + source: a hipModule_t belongs to the device that was current when it was loaded
    std::string text;             // the source alone (band_function compiles more instantiations from it)
    int device = 0;
};
static std::map<std::string, JitModule *> g_modules;

// a compiled program as a loaded module.  bitcode empty: the program's code object; else the program (compiled with -fgpu-rdc) and the
// caller's bitcode linked into one code object first (hiprtcLink*, LLVM bitcode inputs: the caller's row function is inlined into the
// kernels like a source functor's call operator)
static const char *const kJitOpts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-math-errno", "-fgpu-rdc"};
static hipError_t load_program(const Hiprtc *R, hiprtcProgram prog, const std::vector<char> &bitcode, hipModule_t *mod, std::string *why)
{
    if (bitcode.empty()) {
        size_t cs = 0;
        std::vector<char> code;
        if (R->GetCodeSize(prog, &cs) != HIPRTC_SUCCESS || cs == 0) { *why = "hiprtcGetCodeSize failed"; return hipErrorUnknown; }
        code.resize(cs);
        if (R->GetCode(prog, code.data()) != HIPRTC_SUCCESS) { *why = "hiprtcGetCode failed"; return hipErrorUnknown; }
        return hipModuleLoadData(mod, code.data());
    }
    if (!R->GetBitcode || !R->GetBitcodeSize || !R->LinkCreate || !R->LinkAddData || !R->LinkComplete || !R->LinkDestroy) { *why = "this hiprtc has no link interface"; return hipErrorNotSupported; }
    size_t bs = 0;
    if (R->GetBitcodeSize(prog, &bs) != HIPRTC_SUCCESS || bs == 0) { *why = "hiprtcGetBitcodeSize failed"; return hipErrorUnknown; }
    std::vector<char> glue(bs), user(bitcode);
    if (R->GetBitcode(prog, glue.data()) != HIPRTC_SUCCESS) { *why = "hiprtcGetBitcode failed"; return hipErrorUnknown; }
    hiprtcLinkState ls = nullptr;
    if (R->LinkCreate(0, nullptr, nullptr, &ls) != HIPRTC_SUCCESS) { *why = "hiprtcLinkCreate failed"; return hipErrorUnknown; }
    hipError_t e = hipErrorUnknown;
    void *bin = nullptr;
    size_t sz = 0;
    hiprtcResult r = R->LinkAddData(ls, HIPRTC_JIT_INPUT_LLVM_BITCODE, glue.data(), glue.size(), "fdjac kernels", 0, nullptr, nullptr);
    if (r == HIPRTC_SUCCESS) r = R->LinkAddData(ls, HIPRTC_JIT_INPUT_LLVM_BITCODE, user.data(), user.size(), "caller's row function", 0, nullptr, nullptr);
    if (r == HIPRTC_SUCCESS) r = R->LinkComplete(ls, &bin, &sz);
    if (r == HIPRTC_SUCCESS && bin && sz) e = hipModuleLoadData(mod, bin);
    else *why = std::string("linking the caller's bitcode failed (") + R->GetErrorString(r) + "): does it define fdjac_user_row (and fdjac_user_row_c for the complex step) for gfx950?";
    (void)R->LinkDestroy(ls);
    return e;
}

// fd_band_store_cols for the bandwidths (l, u) of this module's functor: the precompiled pair, or one more small compilation
// (the same text, two name expressions) on first use -- about a second, once per functor text and bandwidth pair
static hipFunction_t band_function(JitModule *m, int l, int u, int central)
{
    if (l == 1 && u == 1) return m->band[0][central];
    if (l == 2 && u == 2) return m->band[1][central];
    if (l < 0 || u < 0 || l + u > 8) return nullptr;                  // (2 (l + u + 1) quotients per lane live in registers)
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    JitModule::BandExtra &x = m->extra[std::make_pair(l, u)];
    if (x.fn[central] || x.failed) return x.fn[central];
    x.failed = true;                                                  // (until everything below has worked)
    const Hiprtc *R = hiprtc();
    if (!R) return nullptr;
    hiprtcProgram prog = nullptr;
    if (hipSetDevice(m->device) != hipSuccess) return nullptr;       // (the extra module must live where the functor's first module does)
    if (R->CreateProgram(&prog, m->text.c_str(), "fdjac_jit_band.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return nullptr;
    std::string names[2];
    for (int md = 0; md < 2; ++md) {
        names[md] = "fd_band_store_cols<" + m->real + ", " + (md ? "1" : "0") + ", fdjit_F, " + std::to_string(l) + ", " + std::to_string(u) + ">";
        (void)R->AddNameExpression(prog, names[md].c_str());
    }
    bool ok = R->CompileProgram(prog, m->bitcode.empty() ? 5 : 6, kJitOpts) == HIPRTC_SUCCESS;
    std::string low[2];
    for (int md = 0; md < 2 && ok; ++md) {
        const char *ln = nullptr;
        if (R->GetLoweredName(prog, names[md].c_str(), &ln) == HIPRTC_SUCCESS && ln) low[md] = ln; else ok = false;
    }
    std::string why;
    if (ok) ok = load_program(R, prog, m->bitcode, &x.mod, &why) == hipSuccess;
    (void)R->DestroyProgram(&prog);
    if (!ok) { (void)hipGetLastError(); return nullptr; }
    for (int md = 0; md < 2; ++md)
        if (hipModuleGetFunction(&x.fn[md], x.mod, low[md].c_str()) != hipSuccess) { x.fn[md] = nullptr; ok = false; }
    if (!ok) { (void)hipGetLastError(); return nullptr; }
    x.failed = false;
    return x.fn[central];
}

static const char kJitTailCplx[] = R"FDJIT(
extern "C" __global__ void __launch_bounds__(256) fdjit_rows_cplx(real_t *__restrict__ fx, const real_t *__restrict__ x, fdjit_F f, long long xs,
                                                                  long long fs, long long r0, long long r1)
{
    const long long r = r0 + (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= r1) return;
    const fd_cplx_plain_point<real_t> P = {x + 2 * (long long)blockIdx.y * xs};
    const fd_cplx<real_t> v = f(r, P);
    fx[2 * ((long long)blockIdx.y * fs + r)] = v.re;
    fx[2 * ((long long)blockIdx.y * fs + r) + 1] = v.im;
}
)FDJIT";

// the complex instantiation of the module's functor (one more compilation, kept with the module); nullptr-safe: m->cplx.ok says whether it exists
static bool cplx_functions(JitModule *m)
{
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    JitModule::Cplx &x = m->cplx;
    if (x.tried) return x.ok;
    x.tried = true;
    const Hiprtc *R = hiprtc();
    if (!R || hipSetDevice(m->device) != hipSuccess) return false;
    const std::string src = m->text + kJitTailCplx;
    hiprtcProgram prog = nullptr;
    if (R->CreateProgram(&prog, src.c_str(), "fdjac_jit_cplx.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return false;
    const char *ct[2] = {"unsigned char", "int"};
    std::string names[2], cnames[2];
    for (int cb = 0; cb < 2; ++cb) {
        names[cb] = "fd_csc_store_cols_cplx<" + m->real + ", " + ct[cb] + ", fdjit_F>";
        (void)R->AddNameExpression(prog, names[cb].c_str());
        cnames[cb] = "fd_colrange_store_cols<" + m->real + ", " + ct[cb] + ", 2, fdjit_F>";
        (void)R->AddNameExpression(prog, cnames[cb].c_str());
    }
    bool ok = R->CompileProgram(prog, m->bitcode.empty() ? 5 : 6, kJitOpts) == HIPRTC_SUCCESS;
    size_t ls = 0;
    if (R->GetProgramLogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1) { x.log.resize(ls); (void)R->GetProgramLog(prog, &x.log[0]); }
    std::string low[2], clow[2];
    for (int cb = 0; cb < 2 && ok; ++cb) {
        const char *ln = nullptr;
        if (R->GetLoweredName(prog, names[cb].c_str(), &ln) == HIPRTC_SUCCESS && ln) low[cb] = ln; else ok = false;
        if (ok && R->GetLoweredName(prog, cnames[cb].c_str(), &ln) == HIPRTC_SUCCESS && ln) clow[cb] = ln;
    }
    if (ok) {
        std::string why;
        ok = load_program(R, prog, m->bitcode, &x.mod, &why) == hipSuccess;
        if (!ok) x.log += why;
    }
    (void)R->DestroyProgram(&prog);
    if (!ok) { (void)hipGetLastError(); return false; }
    ok = hipModuleGetFunction(&x.rows, x.mod, "fdjit_rows_cplx") == hipSuccess;
    for (int cb = 0; cb < 2 && ok; ++cb) ok = hipModuleGetFunction(&x.store[cb], x.mod, low[cb].c_str()) == hipSuccess;
    for (int cb = 0; cb < 2 && ok; ++cb)
        if (clow[cb].empty() || hipModuleGetFunction(&x.colrange[cb], x.mod, clow[cb].c_str()) != hipSuccess) x.colrange[cb] = nullptr;      // (an optimisation)
    if (!ok) (void)hipGetLastError();
    x.ok = ok;
    return ok;
}

// fd_colrange_store_cols for forward / central differences (one more small compilation, kept with the module)
static bool colrange_functions(JitModule *m)
{
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    JitModule::ColRange &x = m->cr;
    if (x.tried) return x.ok;
    x.tried = true;
    const Hiprtc *R = hiprtc();
    if (!R || hipSetDevice(m->device) != hipSuccess) return false;
    hiprtcProgram prog = nullptr;
    if (R->CreateProgram(&prog, m->text.c_str(), "fdjac_jit_colrange.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return false;
    const char *ct[2] = {"unsigned char", "int"};
    std::string names[2][2], bnames[2][2];
    for (int cb = 0; cb < 2; ++cb)
        for (int md = 0; md < 2; ++md) {
            names[cb][md] = "fd_colrange_store_cols<" + m->real + ", " + ct[cb] + ", " + (md ? "1" : "0") + ", fdjit_F>";
            (void)R->AddNameExpression(prog, names[cb][md].c_str());
            bnames[cb][md] = "fd_bbb_store_cols<" + m->real + ", " + ct[cb] + ", " + (md ? "1" : "0") + ", fdjit_F>";
            (void)R->AddNameExpression(prog, bnames[cb][md].c_str());
        }
    bool ok = R->CompileProgram(prog, m->bitcode.empty() ? 5 : 6, kJitOpts) == HIPRTC_SUCCESS;
    std::string low[2][2], blow[2][2];
    for (int cb = 0; cb < 2 && ok; ++cb)
        for (int md = 0; md < 2 && ok; ++md) {
            const char *ln = nullptr;
            if (R->GetLoweredName(prog, names[cb][md].c_str(), &ln) == HIPRTC_SUCCESS && ln) low[cb][md] = ln; else ok = false;
            if (ok && R->GetLoweredName(prog, bnames[cb][md].c_str(), &ln) == HIPRTC_SUCCESS && ln) blow[cb][md] = ln; else ok = false;
        }
    std::string why;
    if (ok) ok = load_program(R, prog, m->bitcode, &x.mod, &why) == hipSuccess;
    (void)R->DestroyProgram(&prog);
    for (int cb = 0; cb < 2 && ok; ++cb)
        for (int md = 0; md < 2 && ok; ++md)
            ok = hipModuleGetFunction(&x.fn[cb][md], x.mod, low[cb][md].c_str()) == hipSuccess && hipModuleGetFunction(&x.bbb[cb][md], x.mod, blow[cb][md].c_str()) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    x.ok = ok;
    return ok;
}

}  // namespace fdjac

struct fd_jit_f {
    fd_ctx *ctx = nullptr;
    fdjac::JitModule *m = nullptr;
    int elem_bytes = 8;
    int64_t M = 0, N = 0;
    std::vector<unsigned char> params;      // the functor object, byte for byte (sizeof(F) bytes)
    int64_t launches = 0;
    bool sep = false;                       // fd_f_compile_terms: a separable functor bound to the row lists of one plan
    unsigned long long plan_serial = 0;     //   that plan's serial (fd_csc_store.plan_serial): only there the row-wise store is taken
    void *d_base = nullptr;                 // f(x) of all rows for the column-range store's forward differences (allocated on first use)
    int64_t row_stores = 0, entries = 0;    // launches of the row-wise store; stored entries of that plan's pattern
};

namespace fdjac {

static const char kJitTail[] = R"FDJIT(
typedef FDJIT_FUNCTOR fdjit_F;
struct fdjit_plain {
    typedef real_t value_type;
    const real_t *x;
    __device__ real_t operator()(long long j) const { return x[j]; }
};
extern "C" __global__ void __launch_bounds__(256) fdjit_rows(real_t *__restrict__ fx, const real_t *__restrict__ x, fdjit_F f, long long xs,
                                                             long long fs, long long r0, long long r1)
{
    const long long r = r0 + (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= r1) return;
    const fdjit_plain P = {x + (long long)blockIdx.y * xs};
    fx[(long long)blockIdx.y * fs + r] = f(r, P);
}
extern "C" __device__ __attribute__((used)) const unsigned fdjit_sizeof_f = sizeof(fdjit_F);
extern "C" __device__ __attribute__((used)) const unsigned fdjit_lists_offset = fd_sep_rows_layout<fdjit_F>::lists_offset;
extern "C" __device__ __attribute__((used)) const unsigned fdjit_terms_bytes = fd_sep_rows_layout<fdjit_F>::terms_bytes;
)FDJIT";

static int jit_launch(void *fctx, void *fx, const void *x, int64_t nbatch, int64_t x_stride, int64_t fx_stride, int64_t row_begin,
                      int64_t row_end, int is_complex, void *stream)
{
    fd_jit_f *j = (fd_jit_f *)fctx;
    if (row_end <= row_begin || nbatch < 1) return 0;
    if (is_complex && !cplx_functions(j->m)) {
        set_error("the functor does not compile for the complex step -- write its call operator on `typename P::value_type` (include/fdjac_device.h, "
                  "\"the complex step for row functors\").  Compiler: %.300s", j->m->cplx.log.c_str());
        t_log = j->m->cplx.log;
        return 3;
    }
    long long xs = x_stride, fs = fx_stride, r0 = row_begin, r1 = row_end;      // (complex points: strides and rows in complex elements)
    void *args[] = {&fx, (void *)&x, (void *)j->params.data(), &xs, &fs, &r0, &r1};
    const unsigned gx = (unsigned)((row_end - row_begin + 255) / 256);
    if (hipModuleLaunchKernel(is_complex ? j->m->cplx.rows : j->m->rows, gx, (unsigned)nbatch, 1, 256, 1, 1, 0, (hipStream_t)stream, args, nullptr) != hipSuccess) return 4;
    j->launches += 1;
    return 0;
}

// the lazy launcher serves exactly one request: store column by column (forward / central, real); everything else is declined
static int jit_launch_lazy(void *fctx, void *fx, const fd_lazy_points *lp, int64_t fx_stride, int64_t row_begin, int64_t row_end, void *stream)
{
    (void)fx; (void)fx_stride; (void)row_begin; (void)row_end;
    fd_jit_f *j = (fd_jit_f *)fctx;
    if (lp->store && lp->store_kind == FD_STORE_BAND && !lp->is_complex) {
        // an exact band with cyclic colours (CSC nzval, BandedMatrix data, Tridiagonal diagonals): fd_band_store_cols -- no index reads;
        // bandwidths (1, 1) and (2, 2) are compiled with the functor, others (l + u <= 8) on first use; wider bands are declined (a CSC
        // pattern then takes the column store)
        fd_band_store bs = *(const fd_band_store *)lp->store;
        const int central = lp->pts == 2 ? 1 : 0;
        if (bs.elem_bytes != j->elem_bytes || bs.M != j->M || bs.N != j->N || bs.col_end <= bs.col_begin || lp->c_lo != 0 || lp->ncolors != bs.C ||
            !(central || (lp->pts == 1 && lp->diff == 2)))
            return FD_LAZY_DECLINED;
        hipFunction_t bf = band_function(j->m, bs.l, bs.u, central);
        if (!bf) return FD_LAZY_DECLINED;
        long long jstart = bs.col_begin & ~1ll;
        const void *x = lp->x, *eps = lp->eps;
        void *args[] = {(void *)j->params.data(), (void *)&x, (void *)&eps, &bs, &jstart};
        const unsigned g = (unsigned)((bs.col_end - jstart + 511) / 512);
        if (hipModuleLaunchKernel(bf, g, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, args, nullptr) != hipSuccess) return 4;
        j->launches += 1;
        return 0;
    }
    if (lp->store && lp->store_kind == FD_STORE_BBB && !lp->is_complex) {
        // BandedBlockBandedMatrix data with uniform blocks: one thread per (column, slot), fd_bbb_store_cols -- every slot of every in-band
        // slab; forward differences take f(x) from one plain evaluation into the functor's own buffer
        fd_bbb_store bb = *(const fd_bbb_store *)lp->store;
        const int central = lp->pts == 2 ? 1 : 0, cbi = bb.color_bytes == 4 ? 1 : 0;
        if (bb.elem_bytes != j->elem_bytes || (bb.color_bytes != 1 && bb.color_bytes != 4) || bb.N != j->N || j->M != j->N || bb.block_size < 1 || lp->c_lo != 0 ||
            lp->ncolors != bb.C || !(central || (lp->pts == 1 && lp->diff == 2)) || !colrange_functions(j->m) || !j->m->cr.bbb[cbi][central])
            return FD_LAZY_DECLINED;
        const long long slots = bb.N * (bb.bl + bb.bu + 1) * (bb.lam + bb.mu + 1);
        if (slots <= 0 || (slots + 255) / 256 >= ((long long)1 << 31)) return FD_LAZY_DECLINED;
        const void *xq = lp->x, *eq = lp->eps, *base = nullptr;
        if (!central) {
            if (!j->d_base && hipMalloc(&j->d_base, (size_t)j->M * (size_t)j->elem_bytes) != hipSuccess) { (void)hipGetLastError(); j->d_base = nullptr; }
            if (j->d_base) {
                void *fxp = j->d_base;
                long long xs = 0, fs = 0, r0 = 0, r1 = j->M;
                void *ra[] = {&fxp, (void *)&xq, (void *)j->params.data(), &xs, &fs, &r0, &r1};
                if (hipModuleLaunchKernel(j->m->rows, (unsigned)((j->M + 255) / 256), 1, 1, 256, 1, 1, 0, (hipStream_t)stream, ra, nullptr) != hipSuccess) return 4;
                j->launches += 1;
                base = j->d_base;
            }
        }
        void *args[] = {(void *)j->params.data(), (void *)&xq, (void *)&eq, &bb, (void *)&base};
        if (hipModuleLaunchKernel(j->m->cr.bbb[cbi][central], (unsigned)((slots + 255) / 256), 1, 1, 256, 1, 1, 0, (hipStream_t)stream, args, nullptr) != hipSuccess) return 4;
        j->launches += 1;
        return 0;
    }
    if (lp->store && lp->store_kind == FD_STORE_COLRANGE) {
        // column-range storage (BlockBandedMatrix data): eight columns per workgroup, their rows dealt out flat (fd_colrange_store_cols);
        // forward / central differences (FD_LAZY_CAP_STORE_COLRANGE: the launch forms f(x) of its rows itself) and the complex step
        fd_colrange_store cr = *(const fd_colrange_store *)lp->store;
        const int central = lp->pts == 2 ? 1 : 0, cbi = cr.color_bytes == 4 ? 1 : 0;
        if (cr.elem_bytes != j->elem_bytes || (cr.color_bytes != 1 && cr.color_bytes != 4) || cr.M != j->M || cr.N != j->N || cr.col_end <= cr.col_begin ||
            !(lp->is_complex || central || (lp->pts == 1 && lp->diff == 2)))
            return FD_LAZY_DECLINED;
        hipFunction_t fn = nullptr;
        if (lp->is_complex) { if (cplx_functions(j->m)) fn = j->m->cplx.colrange[cbi]; }
        else if (colrange_functions(j->m)) fn = j->m->cr.fn[cbi][central];
        if (!fn) return FD_LAZY_DECLINED;
        int c_lo = lp->c_lo, c_hi = lp->c_lo + lp->ncolors;
        const void *xq = lp->x, *eq = lp->eps;
        const void *base = nullptr;
        if (!lp->is_complex && !central) {
            // forward differences: f(x) of all rows ONCE (the plain launcher's kernel into the functor's own buffer) -- the columns of
            // a dense block share their rows, forming f(x) inside the storing launch would double the row evaluations
            if (!j->d_base && hipMalloc(&j->d_base, (size_t)j->M * (size_t)j->elem_bytes) != hipSuccess) { (void)hipGetLastError(); j->d_base = nullptr; }
            if (j->d_base) {
                void *fxp = j->d_base;
                long long xs = 0, fs = 0, r0 = 0, r1 = j->M;
                void *ra[] = {&fxp, (void *)&xq, (void *)j->params.data(), &xs, &fs, &r0, &r1};
                if (hipModuleLaunchKernel(j->m->rows, (unsigned)((j->M + 255) / 256), 1, 1, 256, 1, 1, 0, (hipStream_t)stream, ra, nullptr) != hipSuccess) return 4;
                j->launches += 1;
                base = j->d_base;
            }
        }
        void *args[] = {(void *)j->params.data(), (void *)&xq, (void *)&eq, &c_lo, &c_hi, &cr, (void *)&base};
        const long long nw = (cr.col_end - cr.col_begin + 7) / 8;
        if (hipModuleLaunchKernel(fn, (unsigned)(8 * ((nw + 7) / 8)), 1, 1, 256, 1, 1, 0, (hipStream_t)stream, args, nullptr) != hipSuccess) return 4;
        j->launches += 1;
        return 0;
    }
    if (!lp->store || lp->store_kind != FD_STORE_CSC) return FD_LAZY_DECLINED;
    if (lp->is_complex) {
        // the complex step through the column store (FD_LAZY_CAP_STORE_CSC_COMPLEX): every stored entry's row at x + i eps e_j, imag / eps
        fd_csc_store stc = *(const fd_csc_store *)lp->store;
        if (stc.elem_bytes != j->elem_bytes || (stc.color_bytes != 1 && stc.color_bytes != 4) || stc.M != j->M || stc.N != j->N || stc.col_end <= stc.col_begin)
            return FD_LAZY_DECLINED;
        if (!cplx_functions(j->m)) return FD_LAZY_DECLINED;      // (the hand-over path's plain launcher then reports why)
        const long long nb = (stc.col_end - stc.col_begin + 255) / 256;
        int c_lo = lp->c_lo, c_hi = lp->c_lo + lp->ncolors;
        const void *xq = lp->x, *eq = lp->eps;
        void *args[] = {(void *)j->params.data(), (void *)&xq, (void *)&eq, &c_lo, &c_hi, &stc};
        if (hipModuleLaunchKernel(j->m->cplx.store[stc.color_bytes == 4 ? 1 : 0], (unsigned)(8 * ((nb + 7) / 8)), 1, 1, 256, 1, 1, 0, (hipStream_t)stream, args,
                                  nullptr) != hipSuccess) return 4;
        j->launches += 1;
        return 0;
    }
    fd_csc_store st = *(const fd_csc_store *)lp->store;
    if (st.elem_bytes != j->elem_bytes || (st.color_bytes != 1 && st.color_bytes != 4) || st.M != j->M || st.N != j->N || st.col_end <= st.col_begin)
        return FD_LAZY_DECLINED;
    const long long nblk = (st.col_end - st.col_begin + 255) / 256;
    const unsigned g = (unsigned)(8 * ((nblk + 7) / 8));
    int c_lo = lp->c_lo, c_hi = lp->c_lo + lp->ncolors;
    const void *x = lp->x, *eps = lp->eps;
    const int cb = st.color_bytes == 4 ? 1 : 0, central = lp->pts == 2 ? 1 : 0;
    const int64_t reach = st.reach;
    // a separable functor on the plan its lists came from: the Jacobian row by row (fd_csc_store_rows) -- verified colouring, a locally
    // banded square pattern, every column local; anything else takes the column kernels below (same bits)
    if (j->sep && st.row_ptr && st.row_pack && st.row_tile && st.plan_serial == j->plan_serial && st.valid_coloring && reach > 0 && reach <= 700 && st.M == st.N && st.N >= 2 &&
        st.col_begin == 0 && st.col_end == st.N && j->m->store_rows[cb][central]) {
        // (the tile's share of the lists kept in LDS: the mean row length x 1.25 + slack; longer tiles read the rest from memory)
        const double per_row = (double)j->entries / (double)st.M;
        int cap = (int)std::min<int64_t>((int64_t)(256 * per_row * 1.25) + 64, 3072);
        int ireach = (int)reach;
        // a thread per entry where the plan has the entries in its tiles' row order (fd_csc_store_ents; FDJAC_ROWS_ENTS=1; default: a thread per row)
        const size_t lds_e = j->elem_bytes == 8 ? fd_csc_ents_lds_bytes<double>(reach, lp->ncolors, st.ent_tile_max) : fd_csc_ents_lds_bytes<float>(reach, lp->ncolors, st.ent_tile_max);
        const char *sw = test_switch("FDJAC_ROWS_ENTS");
        if (st.ent_col && st.ent_slot && st.ent_info && j->m->store_ents[cb][central] && lds_e <= 64 * 1024 && (sw && *sw == '1')) {
            void *args[] = {(void *)j->params.data(), (void *)&x, (void *)&eps, &c_lo, &c_hi, &st, &ireach};
            const unsigned gr = (unsigned)(8 * (((st.M + 255) / 256 + 7) / 8));
            if (hipModuleLaunchKernel(j->m->store_ents[cb][central], gr, 1, 1, 256, 1, 1, (unsigned)lds_e, (hipStream_t)stream, args, nullptr) != hipSuccess) return 4;
            j->launches += 1;
            j->row_stores += 1;
            return 0;
        }
        const size_t lds_r = j->elem_bytes == 8 ? fd_csc_rows_lds_bytes<double>(reach, lp->ncolors, cap) : fd_csc_rows_lds_bytes<float>(reach, lp->ncolors, cap);
        if (lds_r <= 64 * 1024) {
            void *args[] = {(void *)j->params.data(), (void *)&x, (void *)&eps, &c_lo, &c_hi, &st, &ireach, &cap};
            const unsigned gr = (unsigned)(8 * (((st.M + 255) / 256 + 7) / 8));
            if (hipModuleLaunchKernel(j->m->store_rows[cb][central], gr, 1, 1, 256, 1, 1, (unsigned)lds_r, (hipStream_t)stream, args, nullptr) != hipSuccess) return 4;
            j->launches += 1;
            j->row_stores += 1;
            return 0;
        }
    }
    // a locally banded pattern with a verified colouring: the workgroup's window of x (and f(x)) in LDS, fd_csc_store_cols_win
    const bool wb = !central && st.fx_base != nullptr;
    const size_t lds = !(reach > 0 && reach <= 700) ? 0 : j->elem_bytes == 8 ? fd_csc_win_lds_bytes<double>(reach, wb) : fd_csc_win_lds_bytes<float>(reach, wb);
    if (st.valid_coloring && lds > 0 && lds <= 64 * 1024 && st.M == st.N && j->m->store_win[cb][central]) {
        int ireach = (int)reach, sb = 0, cap = 0;
        void *args[] = {(void *)j->params.data(), (void *)&x, (void *)&eps, &c_lo, &c_hi, &st, &ireach, &sb, &cap};
        if (hipModuleLaunchKernel(j->m->store_win[cb][central], g, 1, 1, 256, 1, 1, (unsigned)lds, (hipStream_t)stream, args, nullptr) != hipSuccess) return 4;
    } else {
        void *args[] = {(void *)j->params.data(), (void *)&x, (void *)&eps, &c_lo, &c_hi, &st};
        if (hipModuleLaunchKernel(j->m->store[cb][central], g, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, args, nullptr) != hipSuccess) return 4;
    }
    j->launches += 1;
    return 0;
}

static void release_module(JitModule *m)
{
    if (!m) return;
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    if (--m->refs > 0) return;
    g_modules.erase(m->key);
    for (auto &kv : m->extra)
        if (kv.second.mod) (void)hipModuleUnload(kv.second.mod);
    if (m->cplx.mod) (void)hipModuleUnload(m->cplx.mod);
    if (m->cr.mod) (void)hipModuleUnload(m->cr.mod);
    if (m->mod) (void)hipModuleUnload(m->mod);
    delete m;
}

}  // namespace fdjac

using namespace fdjac;

extern "C" {

const char *fd_f_compile_log(void) { return t_log.c_str(); }

// the shared back half of fd_f_compile_rows / fd_f_link_rows_bitcode: `src` is the complete translation unit (device header, element
// type, the functor -- source text, or the shim around the caller's bitcode --, the kernels), `bitcode` the caller's LLVM bitcode or empty
static int jit_build(fd_ctx *ctx, const std::string &src, const std::vector<char> &bitcode, const char *real, const char *functor, const void *params,
                     int64_t params_bytes, int64_t M, int64_t N, int elem_bytes, fd_f_launch *fn_out, fd_f_launch_lazy *lazy_out, int *lazy_caps_out,
                     void **fctx_out, const void *const *sep_lists = nullptr, unsigned long long sep_serial = 0)
{
    const bool sep = sep_lists != nullptr;      // fd_f_compile_terms: FDJIT_FUNCTOR is fd_sep_rows<terms>, sep_lists = {row_ptr, row_col} on the device
    // (modules are per DEVICE: a second context on another GPU compiling the same text must not get device 0's functions)
    const std::string key = std::to_string(ctx->device) + "\n" + src + std::string(bitcode.begin(), bitcode.end());
    JitModule *m = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_jit_mutex);
        auto it = g_modules.find(key);
        if (it != g_modules.end()) { m = it->second; m->refs += 1; }
    }
    if (!m) {
        const Hiprtc *R = nullptr;
        {
            std::lock_guard<std::mutex> lock(g_jit_mutex);
            R = hiprtc();
        }
        if (!R) return FD_ERR_UNSUPPORTED;
        hiprtcProgram prog = nullptr;
        hiprtcResult rr = R->CreateProgram(&prog, src.c_str(), "fdjac_jit.hip", 0, nullptr, nullptr);
        FD_REQUIRE(rr == HIPRTC_SUCCESS, FD_ERR_HIP, "hiprtcCreateProgram failed: %s", R->GetErrorString(rr));
        // the kernels of include/fdjac_device.h instantiated for this functor, found by their lowered names
        std::string names[2][2][2];
        const char *ct[2] = {"unsigned char", "int"};
        for (int w = 0; w < 2; ++w)
            for (int cb = 0; cb < 2; ++cb)
                for (int md = 0; md < 2; ++md) {
                    names[w][cb][md] = std::string(w ? "fd_csc_store_cols_win<" : "fd_csc_store_cols<") + real + ", " + ct[cb] + ", " + (md ? "1" : "0") + ", fdjit_F>";
                    (void)R->AddNameExpression(prog, names[w][cb][md].c_str());
                }
        std::string bnames[2][2];
        for (int wi = 0; wi < 2; ++wi)
            for (int md = 0; md < 2; ++md) {
                bnames[wi][md] = std::string("fd_band_store_cols<") + real + ", " + (md ? "1" : "0") + ", fdjit_F, " + (wi ? "2, 2>" : "1, 1>");
                (void)R->AddNameExpression(prog, bnames[wi][md].c_str());
            }
        std::string rnames[2][2], enames[2][2];
        if (sep)
            for (int cb = 0; cb < 2; ++cb)
                for (int md = 0; md < 2; ++md) {
                    rnames[cb][md] = std::string("fd_csc_store_rows<") + real + ", " + ct[cb] + ", " + (md ? "1" : "0") + ", fdjit_F>";
                    (void)R->AddNameExpression(prog, rnames[cb][md].c_str());
                    enames[cb][md] = std::string("fd_csc_store_ents<") + real + ", " + ct[cb] + ", " + (md ? "1" : "0") + ", fdjit_F>";
                    (void)R->AddNameExpression(prog, enames[cb][md].c_str());
                }
        rr = R->CompileProgram(prog, bitcode.empty() ? 5 : 6, kJitOpts);
        size_t ls = 0;
        if (R->GetProgramLogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
            t_log.resize(ls);
            (void)R->GetProgramLog(prog, &t_log[0]);
        }
        if (rr != HIPRTC_SUCCESS) {
            set_error("compiling the functor failed (%s); the compiler's messages: fd_f_compile_log().  First lines: %.300s", R->GetErrorString(rr), t_log.c_str());
            (void)R->DestroyProgram(&prog);
            return FD_ERR_ARG;
        }
        std::string low[2][2][2];
        for (int w = 0; w < 2; ++w)
            for (int cb = 0; cb < 2; ++cb)
                for (int md = 0; md < 2; ++md) {
                    const char *ln = nullptr;
                    if (R->GetLoweredName(prog, names[w][cb][md].c_str(), &ln) == HIPRTC_SUCCESS && ln) low[w][cb][md] = ln;
                }
        std::string blow[2][2];
        for (int wi = 0; wi < 2; ++wi)
            for (int md = 0; md < 2; ++md) {
                const char *ln = nullptr;
                if (R->GetLoweredName(prog, bnames[wi][md].c_str(), &ln) == HIPRTC_SUCCESS && ln) blow[wi][md] = ln;
            }
        std::string rlow[2][2], elow[2][2];
        if (sep)
            for (int cb = 0; cb < 2; ++cb)
                for (int md = 0; md < 2; ++md) {
                    const char *ln = nullptr;
                    if (R->GetLoweredName(prog, rnames[cb][md].c_str(), &ln) == HIPRTC_SUCCESS && ln) rlow[cb][md] = ln;
                    if (R->GetLoweredName(prog, enames[cb][md].c_str(), &ln) == HIPRTC_SUCCESS && ln) elow[cb][md] = ln;
                }
        m = new (std::nothrow) JitModule();
        if (!m) { (void)R->DestroyProgram(&prog); FD_REQUIRE(false, FD_ERR_NOMEM, "out of host memory"); }
        std::string why;
        hipError_t e = load_program(R, prog, bitcode, &m->mod, &why);
        (void)R->DestroyProgram(&prog);
        if (e != hipSuccess && !why.empty()) {
            set_error("%s", why.c_str());
            t_log += why;
            delete m;
            (void)hipGetLastError();
            return FD_ERR_ARG;
        }
        if (e == hipSuccess) e = hipModuleGetFunction(&m->rows, m->mod, "fdjit_rows");
        for (int cb = 0; cb < 2 && e == hipSuccess; ++cb)
            for (int md = 0; md < 2 && e == hipSuccess; ++md) {
                e = low[0][cb][md].empty() ? hipErrorNotFound : hipModuleGetFunction(&m->store[cb][md], m->mod, low[0][cb][md].c_str());
                if (e == hipSuccess && !low[1][cb][md].empty() && hipModuleGetFunction(&m->store_win[cb][md], m->mod, low[1][cb][md].c_str()) != hipSuccess)
                    m->store_win[cb][md] = nullptr;       // (the windowed form is an optimisation: the plain one serves)
            }
        for (int wi = 0; wi < 2 && e == hipSuccess; ++wi)
            for (int md = 0; md < 2; ++md)
                if (blow[wi][md].empty() || hipModuleGetFunction(&m->band[wi][md], m->mod, blow[wi][md].c_str()) != hipSuccess) m->band[wi][md] = nullptr;      // (an optimisation)
        for (int cb = 0; cb < 2 && e == hipSuccess && sep; ++cb)
            for (int md = 0; md < 2; ++md)
            {
                if (rlow[cb][md].empty() || hipModuleGetFunction(&m->store_rows[cb][md], m->mod, rlow[cb][md].c_str()) != hipSuccess) m->store_rows[cb][md] = nullptr;      // (an optimisation)
                if (elow[cb][md].empty() || hipModuleGetFunction(&m->store_ents[cb][md], m->mod, elow[cb][md].c_str()) != hipSuccess) m->store_ents[cb][md] = nullptr;
            }
        if (e == hipSuccess) {
            hipDeviceptr_t dp = nullptr;
            size_t bytes = 0;
            e = hipModuleGetGlobal(&dp, &bytes, m->mod, "fdjit_sizeof_f");
            if (e == hipSuccess) e = hipMemcpy(&m->sizeof_f, dp, sizeof(unsigned), hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipModuleGetGlobal(&dp, &bytes, m->mod, "fdjit_lists_offset");
            if (e == hipSuccess) e = hipMemcpy(&m->lists_offset, dp, sizeof(unsigned), hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipModuleGetGlobal(&dp, &bytes, m->mod, "fdjit_terms_bytes");
            if (e == hipSuccess) e = hipMemcpy(&m->terms_bytes, dp, sizeof(unsigned), hipMemcpyDeviceToHost);
        }
        if (e != hipSuccess) {
            set_error("loading the compiled functor failed: %s", hipGetErrorString(e));
            if (m->mod) (void)hipModuleUnload(m->mod);
            delete m;
            return FD_ERR_HIP;
        }
        (void)hipGetLastError();
        std::lock_guard<std::mutex> lock(g_jit_mutex);
        auto it = g_modules.find(key);
        if (it != g_modules.end()) {       // (another thread compiled the same text meanwhile: keep theirs)
            (void)hipModuleUnload(m->mod);
            delete m;
            m = it->second;
        } else {
            m->key = key;
            m->text = src;
            m->bitcode = bitcode;
            m->device = ctx->device;
            m->real = real;
            g_modules[key] = m;
        }
        m->refs += 1;
    }
    // an empty functor has sizeof 1; otherwise the caller's bytes ARE the functor object (a separable functor: the TERMS object, the
    // library appends the list pointers where fd_sep_rows keeps them)
    const unsigned want_bytes = sep ? m->terms_bytes : m->sizeof_f;
    if (sep && (m->lists_offset == 0 || m->lists_offset + 2 * sizeof(void *) > m->sizeof_f)) {
        set_error("internal: fd_sep_rows layout not reported by the compiled module");
        release_module(m);
        return FD_ERR_HIP;
    }
    if (!((params_bytes == 0 && want_bytes == 1) || (int64_t)want_bytes == params_bytes)) {
        set_error("the functor %s is %u bytes, %lld bytes of parameters were given", functor, want_bytes, (long long)params_bytes);
        release_module(m);
        return FD_ERR_ARG;
    }
    fd_jit_f *j = new (std::nothrow) fd_jit_f();
    if (!j) { release_module(m); set_error("out of host memory"); return FD_ERR_NOMEM; }
    j->ctx = ctx; j->m = m; j->elem_bytes = elem_bytes; j->M = M; j->N = N;
    j->params.assign(std::max<size_t>(m->sizeof_f, 16), 0);
    if (params_bytes > 0) memcpy(j->params.data(), params, (size_t)params_bytes);
    if (sep) {
        memcpy(j->params.data() + m->lists_offset, &sep_lists[0], sizeof(void *));
        memcpy(j->params.data() + m->lists_offset + sizeof(void *), &sep_lists[1], sizeof(void *));
        j->sep = true;
        j->plan_serial = sep_serial;
        int last = 0;      // (row_ptr[M] = the pattern's stored entries: sizes the row-wise store's staging)
        if (hipMemcpy(&last, (const int *)sep_lists[0] + M, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); last = (int)M; }
        j->entries = last;
    }
    *fn_out = jit_launch;
    if (lazy_out) *lazy_out = jit_launch_lazy;
    // (FD_LAZY_CAP_STORE: exact bands of width (1, 1) / (2, 2) through fd_band_store_cols; every other storing request is declined)
    //  ... and column-range storage through fd_colrange_store_cols, compiled on first use)
    if (lazy_caps_out) *lazy_caps_out = FD_LAZY_CAP_STORE_CSC | FD_LAZY_CAP_STORE_CSC_BASE | FD_LAZY_CAP_STORE_CSC_COMPLEX | FD_LAZY_CAP_STORE | FD_LAZY_CAP_STORE_COLRANGE;
    *fctx_out = j;
    return FD_OK;
}

int fd_f_compile_rows(fd_ctx *ctx, const char *source, const char *functor, const void *params, int64_t params_bytes, int64_t M, int64_t N,
                      int elem_bytes, fd_f_launch *fn_out, fd_f_launch_lazy *lazy_out, int *lazy_caps_out, void **fctx_out)
{
    FD_REQUIRE(ctx && source && functor && fn_out && fctx_out, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(elem_bytes == 8 || elem_bytes == 4, FD_ERR_ARG, "elem_bytes must be 8 (Float64) or 4 (Float32)");
    FD_REQUIRE(M >= 1 && N >= 1, FD_ERR_ARG, "bad shape");
    FD_REQUIRE(params_bytes >= 0 && (params || params_bytes == 0), FD_ERR_ARG, "bad functor parameters");
    for (const char *c = functor; *c; ++c)
        FD_REQUIRE((*c >= 'a' && *c <= 'z') || (*c >= 'A' && *c <= 'Z') || (*c >= '0' && *c <= '9') || *c == '_' || *c == ':' || *c == '<' || *c == '>' || *c == ',' ||
                       *c == ' ',
                   FD_ERR_ARG, "functor must be a type name");
    FD_HIP_CHECK(hipSetDevice(ctx->device));
    t_log.clear();
    const char *real = elem_bytes == 8 ? "double" : "float";
    std::string src;      // (hiprtc declares the HIP runtime itself: no include)
    src += kDeviceHeader;
    src += "\ntypedef ";
    src += real;
    src += " real_t;\n#line 1 \"functor\"\n";
    src += source;
    src += "\n#define FDJIT_FUNCTOR ";
    src += functor;
    src += "\n";
    src += kJitTail;
    return jit_build(ctx, src, std::vector<char>(), real, functor, params, params_bytes, M, N, elem_bytes, fn_out, lazy_out, lazy_caps_out, fctx_out);
}

// A SEPARABLE residual from its term (include/fdjac.h): the terms type wrapped into fd_sep_rows (include/fdjac_device.h), which reads
// the pattern by rows from the lists of the plan the functor is made for.
int fd_f_compile_terms(fd_ctx *ctx, const char *source, const char *terms, const void *params, int64_t params_bytes, int64_t M, int64_t N,
                       int elem_bytes, const void *row_ptr_dev, const void *row_col_dev, uint64_t plan_serial, fd_f_launch *fn_out,
                       fd_f_launch_lazy *lazy_out, int *lazy_caps_out, void **fctx_out)
{
    FD_REQUIRE(ctx && source && terms && fn_out && fctx_out && row_ptr_dev && row_col_dev, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(elem_bytes == 8 || elem_bytes == 4, FD_ERR_ARG, "elem_bytes must be 8 (Float64) or 4 (Float32)");
    FD_REQUIRE(M >= 1 && N >= 1 && M < ((int64_t)1 << 31), FD_ERR_ARG, "bad shape");
    FD_REQUIRE(params_bytes >= 0 && (params || params_bytes == 0), FD_ERR_ARG, "bad functor parameters");
    for (const char *c = terms; *c; ++c)
        FD_REQUIRE((*c >= 'a' && *c <= 'z') || (*c >= 'A' && *c <= 'Z') || (*c >= '0' && *c <= '9') || *c == '_' || *c == ':' || *c == '<' || *c == '>' || *c == ',' ||
                       *c == ' ',
                   FD_ERR_ARG, "terms_type must be a type name");
    FD_HIP_CHECK(hipSetDevice(ctx->device));
    t_log.clear();
    const char *real = elem_bytes == 8 ? "double" : "float";
    std::string src;
    src += kDeviceHeader;
    src += "\ntypedef ";
    src += real;
    src += " real_t;\n#line 1 \"functor\"\n";
    src += source;
    src += "\n#define FDJIT_FUNCTOR fd_sep_rows<";
    src += terms;
    src += " >\n";
    src += kJitTail;
    const void *lists[2] = {row_ptr_dev, row_col_dev};
    const std::string name = std::string("fd_sep_rows<") + terms + " >";
    return jit_build(ctx, src, std::vector<char>(), real, name.c_str(), params, params_bytes, M, N, elem_bytes, fn_out, lazy_out, lazy_caps_out, fctx_out, lists,
                     (unsigned long long)plan_serial);
}


// ---- a row function given as LLVM BITCODE (fd_f_link_rows_bitcode) ------------------------------------------------------------------
// What a caller without C++ hands over: AMDGPU.jl / GPUCompiler emit bitcode for a Julia closure, `hipcc -emit-llvm --offload-device-only
// -fgpu-rdc -c` for a C function.  The bitcode defines, for the element type T of the call,
//     extern "C" __device__ T    fdjac_user_row  (const void *params, long long r, const fd_cpoint *X);                 (required)
//     extern "C" __device__ void fdjac_user_row_c(const void *params, long long r, const fd_cpoint *X, T *re_im);     (complex step: optional)
// and reads the point through what THIS side of the link defines:
//     extern "C" __device__ T    fdjac_point_get  (const fd_cpoint *X, long long j);
//     extern "C" __device__ void fdjac_point_get_c(const fd_cpoint *X, long long j, T *re_im);
// fd_cpoint is {int kind; const void *obj;}: which of the kernels' point types it stands for.  After the link-time inlining the kind is
// a constant in every kernel and the switch is gone: the kernels are those of a source functor.
static const char kExternHead[] = R"FDJIT(
struct fd_cpoint { int kind; const void *obj; };
struct fdjit_plain;
template <class P> struct fd_cpoint_kind;
template <> struct fd_cpoint_kind<fd_column_point<real_t>> { static constexpr int value = 0; };
template <> struct fd_cpoint_kind<fd_colour_point<real_t, unsigned char>> { static constexpr int value = 1; };
template <> struct fd_cpoint_kind<fd_colour_point<real_t, int>> { static constexpr int value = 2; };
template <> struct fd_cpoint_kind<fd_window_column_point<real_t>> { static constexpr int value = 3; };
template <> struct fd_cpoint_kind<fdjit_plain> { static constexpr int value = 4; };
template <> struct fd_cpoint_kind<fd_cplx_column_point<real_t>> { static constexpr int value = 5; };
template <> struct fd_cpoint_kind<fd_cplx_colour_point<real_t, unsigned char>> { static constexpr int value = 6; };
template <> struct fd_cpoint_kind<fd_cplx_colour_point<real_t, int>> { static constexpr int value = 7; };
template <> struct fd_cpoint_kind<fd_cplx_plain_point<real_t>> { static constexpr int value = 8; };
extern "C" __device__ real_t fdjac_user_row(const void *params, long long r, const fd_cpoint *X);
extern "C" __device__ void fdjac_user_row_c(const void *params, long long r, const fd_cpoint *X, real_t *re_im);
struct fd_extern_F {
    unsigned char params[FDJIT_NPARAMS];
    template <class P> __device__ real_t call(long long r, const P &X, real_t *) const
    {
        const fd_cpoint c = {fd_cpoint_kind<P>::value, &X};
        return fdjac_user_row(params, r, &c);
    }
    template <class P> __device__ fd_cplx<real_t> call(long long r, const P &X, fd_cplx<real_t> *) const
    {
        const fd_cpoint c = {fd_cpoint_kind<P>::value, &X};
        real_t o[2];
        fdjac_user_row_c(params, r, &c, o);
        return fd_cplx<real_t>{o[0], o[1]};
    }
    template <class P> __device__ typename P::value_type operator()(long long r, const P &X) const { return call(r, X, (typename P::value_type *)nullptr); }
};
)FDJIT";
static const char kExternTail[] = R"FDJIT(
extern "C" __device__ real_t fdjac_point_get(const fd_cpoint *X, long long j)
{
    switch (X->kind) {
    case 0: return (*(const fd_column_point<real_t> *)X->obj)(j);
    case 1: return (*(const fd_colour_point<real_t, unsigned char> *)X->obj)(j);
    case 2: return (*(const fd_colour_point<real_t, int> *)X->obj)(j);
    case 3: return (*(const fd_window_column_point<real_t> *)X->obj)(j);
    default: return (*(const fdjit_plain *)X->obj)(j);
    }
}
extern "C" __device__ void fdjac_point_get_c(const fd_cpoint *X, long long j, real_t *re_im)
{
    fd_cplx<real_t> v;
    switch (X->kind) {
    case 5: v = (*(const fd_cplx_column_point<real_t> *)X->obj)(j); break;
    case 6: v = (*(const fd_cplx_colour_point<real_t, unsigned char> *)X->obj)(j); break;
    case 7: v = (*(const fd_cplx_colour_point<real_t, int> *)X->obj)(j); break;
    default: v = (*(const fd_cplx_plain_point<real_t> *)X->obj)(j); break;
    }
    re_im[0] = v.re;
    re_im[1] = v.im;
}
)FDJIT";

int fd_f_link_rows_bitcode(fd_ctx *ctx, const void *bitcode, int64_t bitcode_bytes, const void *params, int64_t params_bytes, int64_t M, int64_t N,
                           int elem_bytes, fd_f_launch *fn_out, fd_f_launch_lazy *lazy_out, int *lazy_caps_out, void **fctx_out)
{
    FD_REQUIRE(ctx && bitcode && bitcode_bytes > 0 && fn_out && fctx_out, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(elem_bytes == 8 || elem_bytes == 4, FD_ERR_ARG, "elem_bytes must be 8 (Float64) or 4 (Float32)");
    FD_REQUIRE(M >= 1 && N >= 1, FD_ERR_ARG, "bad shape");
    FD_REQUIRE(params_bytes >= 0 && params_bytes <= 4096 && (params || params_bytes == 0), FD_ERR_ARG, "bad parameters (at most 4096 bytes: they travel as a kernel argument)");
    FD_HIP_CHECK(hipSetDevice(ctx->device));
    t_log.clear();
    const char *real = elem_bytes == 8 ? "double" : "float";
    const int64_t np = params_bytes > 0 ? params_bytes : 1;
    std::string src;
    src += kDeviceHeader;
    src += "\ntypedef ";
    src += real;
    src += " real_t;\n#define FDJIT_NPARAMS " + std::to_string((long long)np) + "\n";
    src += kExternHead;
    src += "\n#define FDJIT_FUNCTOR fd_extern_F\n";
    src += kJitTail;
    src += kExternTail;
    const std::vector<char> bc((const char *)bitcode, (const char *)bitcode + bitcode_bytes);
    // (an empty parameter block is one byte of padding: the functor object cannot be empty)
    const unsigned char zero = 0;
    return jit_build(ctx, src, bc, real, "fd_extern_F", params_bytes > 0 ? params : &zero, np, M, N, elem_bytes, fn_out, lazy_out, lazy_caps_out, fctx_out);
}

int fd_f_compiled_destroy(void *fctx)
{
    fd_jit_f *j = (fd_jit_f *)fctx;
    if (!j) return FD_OK;
    (void)hipSetDevice(j->ctx->device);
    (void)hipStreamSynchronize(j->ctx->stream);
    release_module(j->m);
    if (j->d_base) (void)hipFree(j->d_base);
    delete j;
    return FD_OK;
}

int fd_f_compiled_counts(void *fctx, int64_t *launches)
{
    FD_REQUIRE(fctx && launches, FD_ERR_ARG, "NULL argument");
    *launches = ((fd_jit_f *)fctx)->launches;
    return FD_OK;
}

int fd_f_compiled_row_stores(void *fctx, int64_t *row_stores)
{
    FD_REQUIRE(fctx && row_stores, FD_ERR_ARG, "NULL argument");
    *row_stores = ((fd_jit_f *)fctx)->row_stores;
    return FD_OK;
}

}  // extern "C"

#endif /* FDJAC_F32 */
