// kf_xengine_q1.hip -- the XCD-confined decode engines on 1-bit and 2-bit PackedQ layers (round 6; 1-bit: BASELINE config 5's storage): the kernel of kf_xengine_kernel.h
// instantiated for FMT_Q1T / FMT_Q2T -- a lane takes ONE DWORD of a 128-element 1-bit block, resp. one 8-byte half of a 64-element 2-bit block (32 weights either way: the
// lanes, slots and canonical chains of a 4-bit matrix, kf_engine.hip's dealing), BlockPrep<FMT> turns it into 16 bf16 pair words through the 256-entry LDS selector table, and
// every sequence of the decoder takes its chain pair over them.  A translation unit of its own, so that the storages' instantiations compile side by side.
#include "kf_xengine_kernel.h"

namespace kf {

template <int FMT, int NWV, int DEPTH, int NB>
using XL1 = XCfg<FMT, 2, 128, NWV, 1024, 2048, 1024, 3072, DEPTH, false, 1, 2, false, NB>; /* Qwen3-0.6B */
template <int FMT, int NWV, int DEPTH, int NB>
using XL2 = XCfg<FMT, 2, 64, NWV, 256, 256, 128, 512, DEPTH, false, 1, 2, false, NB>;       /* the 256-wide test shape */

// one / two / four sequences per decoder (n_seq <= 8 / 16 / 32), the default forms of the 4-bit engines
template <int FMT>
static int go_fmt(XEngineHost* E, hipStream_t st) {
    const int n = E->args.n_seq, nb = n <= XE_NXCD ? 1 : (n <= 2 * XE_NXCD ? 2 : 4);
    if (E->shape_class == 1) {
        if (nb == 4) return xengine_go<XL1<FMT, 12, 2, 4>>(E, st);
        if (nb == 2) return xengine_go<XL1<FMT, 12, 4, 2>>(E, st);
        return xengine_go<XL1<FMT, 12, 2, 1>>(E, st);
    }
    if (E->shape_class == 2) {
        if (nb == 4) return xengine_go<XL2<FMT, 12, 2, 4>>(E, st);
        if (nb == 2) return xengine_go<XL2<FMT, 12, 4, 2>>(E, st);
        return xengine_go<XL2<FMT, 12, 2, 1>>(E, st);
    }
    return KF_UNSUPPORTED_DATATYPE;
}
template <int FMT>
static size_t smem_fmt(int sc, int n_seq, int n_layer) {
    const int nb = n_seq <= XE_NXCD ? 1 : (n_seq <= 2 * XE_NXCD ? 2 : 4);
    if (sc == 1) return nb == 4 ? xe_smem<XL1<FMT, 12, 2, 4>>(n_layer) : (nb == 2 ? xe_smem<XL1<FMT, 12, 4, 2>>(n_layer) : xe_smem<XL1<FMT, 12, 2, 1>>(n_layer));
    return nb == 4 ? xe_smem<XL2<FMT, 12, 2, 4>>(n_layer) : (nb == 2 ? xe_smem<XL2<FMT, 12, 4, 2>>(n_layer) : xe_smem<XL2<FMT, 12, 2, 1>>(n_layer));
}
int xengine_go_lowbit(XEngineHost* E, hipStream_t st) { return E->fmt == FMT_Q1T ? go_fmt<FMT_Q1T>(E, st) : (E->fmt == FMT_Q2T ? go_fmt<FMT_Q2T>(E, st) : KF_UNSUPPORTED_DATATYPE); }
size_t xe_smem_lowbit(int fmt, int sc, int n_seq, int n_layer) { return fmt == FMT_Q1T ? smem_fmt<FMT_Q1T>(sc, n_seq, n_layer) : smem_fmt<FMT_Q2T>(sc, n_seq, n_layer); }

}  // namespace kf
