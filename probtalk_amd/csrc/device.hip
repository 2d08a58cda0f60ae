// Per-device launch state shared by the kernel launchers (pt_common.h).
#include <atomic>

#include "pt_common.h"

namespace pt {

static int current_device_checked(int* dev) {
  PT_HIP(hipGetDevice(dev));
  PT_CHECK(*dev >= 0 && *dev < PT_MAX_DEVICES, "device ordinal %d out of range (the library keeps state for %d devices)", *dev, PT_MAX_DEVICES);
  return 0;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device): a process that uses two GPUs has to set
// it on each.  `st` is the caller's per-kernel record of the largest size set on each device.  Launchers of one kernel
// may race here from two threads: both then set the same attribute to the same value, which is harmless.
int ensure_dyn_lds(const void* kernel, int bytes, LdsAttr& st) {
  int dev = 0;
  if (current_device_checked(&dev)) return 1;
  if (__atomic_load_n(&st.bytes[dev], __ATOMIC_RELAXED) >= bytes) return 0;
  PT_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  __atomic_store_n(&st.bytes[dev], bytes, __ATOMIC_RELAXED);
  return 0;
}

int device_cus() {
  static int cus[PT_MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PT_MAX_DEVICES) return 256;
  int n = __atomic_load_n(&cus[dev], __ATOMIC_RELAXED);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    __atomic_store_n(&cus[dev], n, __ATOMIC_RELAXED);
  }
  return n;
}

}  // namespace pt
