// ABI bookkeeping for libvidar_hip.so
#include "vidar_hip.h"
extern "C" int vidar_abi_version(void) { return 1; }
