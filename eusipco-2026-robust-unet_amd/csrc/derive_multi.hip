// Every derived weight of a training step in ONE launch.  The split-operand paths read their weights as bf16 planes (F(4x4) and F(2x2)
// Winograd-domain filters, packed 1x1 / transposed-convolution matrices: derive_weights.h), refreshed once per optimizer step - 67 launches of
// 5-30 us each for the Robust U-Net (32 + 8 + 27), most of them far too small to fill 256 CUs.  Here a table in device memory lists the
// tensors with the first block of each; a block finds its entry by bisection (scalar, <= 8 steps) and runs that tensor's body.  The table is
// a function of the parameter / destination ADDRESSES only, so the host builds and uploads it once and replays the launch every step.
#include "derive_weights.h"

namespace {

struct DeriveDesc {          // 64 bytes; host and device agree on this layout, nobody else sees it
    const float* w;
    __bf16* dst;
    long stride_z, sk, sn;   // pack kind only
    int kind;                // RUNET_DERIVE_*
    int a, b, c;             // Winograd kinds: cin, cout, dgrad;  pack kind: batch (taps), k, n
    int first_block, blocks;
};
static_assert(sizeof(DeriveDesc) == 64, "descriptor layout");

__global__ __launch_bounds__(256) void derive_multi_kernel(const DeriveDesc* __restrict__ tab, int n) {
    const int blk = blockIdx.x;
    int lo = 0, hi = n - 1;                  // last entry with first_block <= blk
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    const DeriveDesc d = tab[lo];
    const long vb = blk - d.first_block;
    if (vb >= d.blocks) return;
    switch (d.kind) {
    case RUNET_DERIVE_WINO4: derive::wino4_weight_x3_body(d.w, d.dst, d.a, d.b, d.c, vb); break;
    case RUNET_DERIVE_WINO2: derive::wino_weight_x3_body(d.w, d.dst, d.a, d.b, d.c, vb); break;
    default: derive::pack_x3_body(d.w, d.stride_z, d.sk, d.sn, d.dst, d.a, d.b, d.c, vb); break;
    }
}

}  // namespace

extern "C" int runet_derive_desc_bytes(void) { return (int)sizeof(DeriveDesc); }

// Fills entry `index` of a HOST table (runet_derive_desc_bytes() bytes per entry).  kind RUNET_DERIVE_WINO4 / _WINO2: the arguments of
// runet_wino4_weights_x3 / runet_wino_weights_x3 (cin, cout of the module's weight, mode = dgrad);  RUNET_DERIVE_PACK: those of
// runet_conv_x3_pack (cin = channels READ in `mode`, cout = channels WRITTEN).  first_block: sum of the blocks of the entries before it.
// -> blocks of this entry (> 0), or a negative error code.
extern "C" int runet_derive_desc(void* host_table, int index, int kind, const float* w, void* dst, int cin, int cout, int mode, int first_block) {
    if (!host_table || index < 0 || !w || !dst || cin <= 0 || cout <= 0 || first_block < 0 || ((uintptr_t)dst % 16) != 0) return -1;
    DeriveDesc d{};
    d.w = w; d.dst = (__bf16*)dst; d.kind = kind; d.first_block = first_block;
    long items;
    if (kind == RUNET_DERIVE_WINO4 || kind == RUNET_DERIVE_WINO2) {
        if (mode < 0 || mode > (kind == RUNET_DERIVE_WINO4 ? 2 : 1)) return -1;
        const int k = mode ? cout : cin, n = mode ? cin : cout;
        if (k % 8) return -1;
        d.a = cin; d.b = cout; d.c = mode;
        items = (long)(k / 8) * n;
    } else if (kind == RUNET_DERIVE_PACK) {
        if (!runet_conv_x3_supported(cin, cout, mode)) return -1;
        int taps;
        derive::conv_x3_pack_strides(cin, cout, mode, taps, d.stride_z, d.sk, d.sn);
        d.a = taps; d.b = cin; d.c = cout;
        items = (long)taps * (cin / 8) * cout;
    } else {
        return -1;
    }
    const long blocks = cdiv(items, 256);
    if (blocks > (1L << 24)) return -1;
    d.blocks = (int)blocks;
    reinterpret_cast<DeriveDesc*>(host_table)[index] = d;
    return d.blocks;
}

// table: DEVICE copy of n_desc entries filled by runet_derive_desc, first_block ascending; total_blocks = sum of their blocks.
extern "C" int runet_derive_multi(const void* table, int n_desc, int total_blocks, void* stream) {
    RUNET_REQUIRE(table && n_desc > 0 && total_blocks > 0, "bad arguments");
    hipLaunchKernelGGL(derive_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const DeriveDesc*)table, n_desc);
    RUNET_CHECK_LAUNCH();
}
