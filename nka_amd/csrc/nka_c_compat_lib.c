/* nka_c_compat_lib.c -> nka_amd/libnka_c_compat.so: the nine functions of the reference's C API
 * (/root/reference/src-C/nonlinear_krylov_accelerator.h:3-12: nka_init, nka_delete, nka_accel_update,
 * nka_restart, nka_relax, nka_num_vec, nka_max_vec, nka_vec_len, nka_vec_tol) as EXPORTED symbols over
 * libnka_hip.so, so that a caller written for the reference links unchanged -- its own include line and
 * the reference's own header included.  The bodies are those of include/nka_c_compat.h. */
#define NKA_C_COMPAT_API
#include "../../include/nka_c_compat.h"
