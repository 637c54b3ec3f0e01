// rccl_dl.hpp -- RCCL bound at first use (dlopen), shared by nka_hip.hip (array flavour:
// one all-reduce of 2+2*mvec doubles per update) and vec_ops.hip (device vector types: the
// reductions of the abstract-vector flavour).
//
// RCCL is bound at first use, not at link time, so that a process holds exactly
// ONE copy of it: if a librccl.so.1 is already mapped (PyTorch ships its own and
// loads it with `import torch`), that copy is used -- two RCCLs in one process
// would each bring their own communicator state and kernels --, otherwise the
// ROCm installation's (library RUNPATH /opt/rocm/lib).  NKA_HIP_RCCL_LIB names
// another file.  Single-GPU users never load the 570 MB library at all.
#pragma once

#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdlib>
#include <string>

namespace nka_detail {

struct Rccl {
  void *handle = nullptr;
  std::string path, err;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  Rccl() {
    const char *user = getenv("NKA_HIP_RCCL_LIB");
    if (user && *user) {
      handle = dlopen(user, RTLD_NOW | RTLD_LOCAL);
    } else {
      handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);   // the copy already in this process
      if (!handle) handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
      if (!handle) handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    }
    if (!handle) {
      const char *e = dlerror();
      err = std::string("cannot load RCCL: ") + (e ? e : "unknown dlopen error");
      return;
    }
#define NKA_SYM(name) name = reinterpret_cast<decltype(name)>(dlsym(handle, "nccl" #name))
    NKA_SYM(GetUniqueId);
    NKA_SYM(CommInitRank);
    NKA_SYM(CommDestroy);
    NKA_SYM(CommCount);
    NKA_SYM(CommUserRank);
    NKA_SYM(AllReduce);
    NKA_SYM(GetErrorString);
#undef NKA_SYM
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !CommCount || !CommUserRank || !AllReduce || !GetErrorString) {
      err = "the loaded RCCL lacks a required ncclXxx symbol";
      handle = nullptr;
      return;
    }
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(AllReduce), &info) && info.dli_fname) path = info.dli_fname;
  }
  bool ok() const { return handle != nullptr; }
};

// one instance per shared object (inline function, static local)
inline const Rccl &rccl() {
  static const Rccl r;
  return r;
}

}  // namespace nka_detail
