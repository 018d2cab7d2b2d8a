// Ablation switches of the measurement build (make ab-lib / dbg-libs, -DFX_AB: the FIAT_AMD_DEBUG bits arrive in the
// `debug` field of the kernel arguments -- 1 skip the recurrence, 2 skip the contraction, 4 skip the HBM stores,
// 8 kernel-specific).  In the product build FX_ABL is the constant 0, so the branches compile away: no
// store-skipping switch is reachable from a struct field of libfiat_amd.so.  (Bits 16-17 of `debug` carry the Piola
// kind of the fused push-forward; they are functional, not ablation, and are read directly.)
#pragma once
#ifdef FX_AB
#define FX_ABL(args, bit) ((args).debug & (bit))
#else
#define FX_ABL(args, bit) 0
#endif
