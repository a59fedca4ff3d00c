// Runtime D -> compile-time (padded dim DP, lanes per matrix GG, waves per block WW).
// D is padded with identity rows/cols up to DP, so every 1 <= D <= 64 is served by 7 instantiations.
#pragma once
#define VBMP_MAX_DIM 64
#define VBMP_ERR_ARG (-1)
#define VBMP_ERR_LAUNCH (-2)

#ifndef VBMP_W16
#define VBMP_W16 4 /* waves per block at 9 <= D <= 16 (A/B builds: tools/exp/build_variant.sh ... -DVBMP_W16=n) */
#endif
#define VBMP_DISPATCH_CASE_(dp, gg, ww, ...) \
  {                                           \
    constexpr int DP = dp, GG = gg, WW = ww;  \
    __VA_ARGS__;                              \
  }

#define VBMP_DISPATCH_DIM(T, D, ...)                      \
  do {                                                     \
    const int d__ = (D);                                   \
    if (d__ <= 1) VBMP_DISPATCH_CASE_(1, 1, 4, __VA_ARGS__) \
    else if (d__ <= 2) VBMP_DISPATCH_CASE_(2, 1, 4, __VA_ARGS__) \
    else if (d__ <= 4) VBMP_DISPATCH_CASE_(4, 1, 4, __VA_ARGS__) \
    else if (d__ <= 8) VBMP_DISPATCH_CASE_(8, 4, 4, __VA_ARGS__) \
    else if (d__ <= 16) VBMP_DISPATCH_CASE_(16, 16, VBMP_W16, __VA_ARGS__) \
    else if (d__ <= 32) VBMP_DISPATCH_CASE_(32, 16, 2, __VA_ARGS__) \
    else if (d__ <= 64) VBMP_DISPATCH_CASE_(64, 64, 1, __VA_ARGS__) \
  } while (0)
