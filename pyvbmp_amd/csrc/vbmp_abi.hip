// ABI version probe (no GPU work).
#include "../../include/vbmp_hip.h"
extern "C" int vbmp_abi_version(void) { return VBMP_ABI_VERSION; }

// tuning switches (bit 0: non-temporal tile loads, bit 1: non-temporal tile stores); not part of the ABI contract
extern "C" int g_vbmp_flags = 0;
extern "C" void vbmp_debug_set_flags(int f) { g_vbmp_flags = f; }
extern "C" int g_vbmp_blocks_per_cu = 0;  // 0 = default persistent grid (8 blocks per CU)
extern "C" void vbmp_debug_set_blocks_per_cu(int n) { g_vbmp_blocks_per_cu = n; }
