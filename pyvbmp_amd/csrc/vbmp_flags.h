// Debug / A-B switches of the library (vbmp_debug_set_flags; NOT part of the C-ABI contract: tests and tools/exp only).
// One bit per switch, defined once, so that forcing one kernel's form never changes another kernel's path.
#pragma once
#define VBMP_DBG_NT_LOAD        0x1        /* K1/K2: non-temporal tile loads */
#define VBMP_DBG_NT_STORE       0x2        /* K1/K2: non-temporal tile stores */
#define VBMP_DBG_PLAIN_ORDER    0x4        /* K1/K2: tiles in plain blockIdx order (round 2) instead of XCD-contiguous */
#define VBMP_DBG_BLK_INV_WAVES  0x40000000 /* K9 block form, -DVBMP_BLK_INV_MFMA builds only: inverses by one wave per matrix instead of the matrix-core block Gauss-Jordan */
#define VBMP_DBG_ROWS_DIRECT     0x8        /* K12 + quad: a row per thread straight from global memory (round 2) instead of rows staged through LDS */
#define VBMP_DBG_LDS_LANES      0x10       /* K9: lane-per-series form */
#define VBMP_DBG_LDS_ROWS       0x20       /* K9: row-per-lane form */
#define VBMP_DBG_K1_WAVE        0x40       /* K1: one wave per matrix */
#define VBMP_DBG_K1_BLOCK       0x80       /* K1: one block per matrix (33 <= D < 64) */
#define VBMP_DBG_QF_VALU        0x100      /* K3/K3a: VALU form instead of the matrix cores */
#define VBMP_DBG_LDS_BLOCK      0x200      /* K9: block-per-series form for any H */
#define VBMP_DBG_K2_SKELETON    0x400      /* K2, -DVBMP_K2_EXP builds only: no elimination (memory skeleton, WRONG results) */
#define VBMP_DBG_FP_EXACT       0x800      /* K9: fixed-point shortcut only at a bitwise repeat (= VBMP_LDS_FIXED_POINT_EXACT) */
#define VBMP_DBG_BLK_INV_WIDE   0x1000     /* K9 block form, -DVBMP_BLK_INV_BLOCKWIDE builds: block-wide inverses */
#define VBMP_DBG_RUNTIME_D      0x2000     /* K1/K2: run-time D instead of the compile-time 6 / 12 / 20 instances */
#define VBMP_DBG_ESTEP_2KERNEL  0x4000     /* K3: likelihood and softmax as two kernels */
#define VBMP_DBG_FP_OFF         0x8000     /* K9: literal recursion at every step (= VBMP_LDS_FIXED_POINT_OFF) */
#define VBMP_DBG_IDLE_LDS_SHIFT 16         /* K2: bits 16-23 = KB of idle dynamic LDS per block (occupancy experiments) */
#define VBMP_DBG_MNW_GENERIC    0x1000000  /* K7/K8: generic instance instead of the direction-specialised ones */
#define VBMP_DBG_BLK_GENERIC    0x2000000  /* K9 block form: run-time-H instance instead of the compile-time ones */
#define VBMP_DBG_BLK_MONO       0x4000000  /* K9 block form: one block per series for the whole sweep (no forward / Gamma-chain split) */
#define VBMP_DBG_MATSUM_VALU    0x8000000  /* K5b: VALU form for any number of weight columns (no matrix-core form) */
#define VBMP_DBG_SCHED_SHIFT    28         /* K1/K2: bits 28-29 = 8 / 16 / 2 tile groups instead of 4 */
