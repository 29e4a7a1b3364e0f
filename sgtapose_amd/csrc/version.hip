// libsgta_hip.so housekeeping entry points.
#include "common.h"

extern "C" int sgta_version(void) { return 100; }   // 0.1.0

// 0 when a gfx950 device is current; >0 hipError_t; -2 when the device is not gfx950.
extern "C" int sgta_check_device(void) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  hipDeviceProp_t p;
  e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) return (int)e;
  const char* a = p.gcnArchName;
  return (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? 0 : -2;
}
