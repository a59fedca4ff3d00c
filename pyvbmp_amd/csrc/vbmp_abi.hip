// ABI version probe (no GPU work).
#include "../../include/vbmp_hip.h"
extern "C" int vbmp_abi_version(void) { return VBMP_ABI_VERSION; }
