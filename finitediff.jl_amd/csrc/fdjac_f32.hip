// The Float32 instantiation of libfdjac (symbols fd32_*, include/fdjac.h): one unity translation unit that re-compiles
// the element-type dependent sources with real_t = float.  fdjac_internal.h renames every element-type dependent symbol; contexts, the
// error text, colouring and the copy probe stay with the Float64 build (their definitions are skipped here).
#define FDJAC_F32 1
#include "fdjac_kernels.hip"
#include "fdjac_api.hip"
#include "fdjac_match.hip"
#include "fdjac_builtin_f.hip"
#include "fdjac_jvp.hip"
#include "fdjac_solve.hip"
#include "fdjac_bandsolve.hip"
#include "fdjac_blocksolve.hip"
