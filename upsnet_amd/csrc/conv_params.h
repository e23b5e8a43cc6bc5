// conv_params.h -- launch description shared by the convolution kernels (conv.hip: fp32 MFMA; conv_bf16.hip: bf16 MFMA).
#pragma once
#include "common.h"

#define CV_BM 128
#define CV_BK 32
#define CV_LDA (CV_BM + 1)
#define CV_MAXSEG 5

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct ConvSeg {
    const float *x, *res, *off, *mask;
    const float *w;   // this map's own packed weights (conv.hip, upsnet_conv2d_nhwc_f32_multiw); NULL: the launch's shared p.w
    float *out;
    int N, H, W, Ho, Wo;
    int OH, OW;  // Winograd: real output size (Ho, Wo then count 2x2 output tiles)
    int tile_start;
    long M;  // N*Ho*Wo
};

struct ConvParams {
    ConvSeg seg[CV_MAXSEG];
    const float *w, *bias;
    int nseg, Cin, Cout, ldw, KH, KW, stride, pad, dil, relu;
    int res_up;  // epilogue mode: 1 = residual at half resolution, added through a nearest x2 upsampling (FPN top-down
                 // path); 2 = 2x2 / stride-2 transposed-convolution scatter
    int m_tiles, n_tiles;
    int ksplit;        // split-K: the K walk is divided over `ksplit` workgroups per tile; each writes raw partial sums
    float *partial;    // [ksplit][sum of M over the maps][Cout] (workspace), reduced by conv_splitk_reduce_kernel
    long m_total;      // sum of M over the maps
    int io;            // conv_bf16.hip: bit 0 / 1 / 2 = the input / output / residual tensors are bf16 instead of fp32
    // conv1x1.hip, sibling launch: output channels [sib_split, Cout) belong to a second layer on the same input -- they go to sib_out (an
    // [M, Cout - sib_split] tensor) with ReLU flag sib_relu; channels [0, sib_split) to seg[0].out ([M, sib_split]) with p.relu. 0: one layer
    int sib_split, sib_relu;
    float *sib_out;
};

// Validates the arguments of a convolution entry point and fills the per-map descriptors (output geometry, pixel counts).
static inline int conv_fill(ConvParams &p, const char *who, int nseg, const float *const x[], const float *const res[],
                     const float *const off[], const float *const mask[], float *const out[], const int batch[],
                     const int height[], const int width[], int Cin, int Cout, const float *wpack, int ldw, const float *bias,
                     int KH, int KW, int stride, int pad, int dil, int relu)
{
    UPS_REQUIRE(nseg >= 1 && nseg <= CV_MAXSEG, "%s: 1..%d feature maps per launch (got %d)", who, CV_MAXSEG, nseg);
    UPS_REQUIRE(x && out && height && width && wpack, "%s: null pointer", who);
    UPS_REQUIRE(Cin > 0 && Cin % CV_BK == 0, "%s: Cin must be a multiple of 32 (got %d)", who, Cin);
    UPS_REQUIRE(Cout > 0 && ldw % 32 == 0 && ldw >= Cout, "%s: ldw must be Cout rounded up to 32 (got %d for Cout=%d)", who, ldw, Cout);
    UPS_REQUIRE(KH >= 1 && KW >= 1 && KH * KW <= 49 && stride >= 1 && pad >= 0 && dil >= 1, "%s: bad kernel/stride/pad/dilation", who);
    UPS_REQUIRE((long)KH * KW * Cin * ldw < (1L << 30), "%s: packed weight exceeds 4 GiB", who);
    p.w = wpack; p.bias = bias; p.nseg = nseg; p.Cin = Cin; p.Cout = Cout; p.ldw = ldw; p.KH = KH; p.KW = KW;
    p.stride = stride; p.pad = pad; p.dil = dil; p.relu = relu; p.res_up = 0;
    p.ksplit = 1; p.partial = nullptr; p.m_total = 0; p.io = 0; p.sib_split = 0; p.sib_relu = 0; p.sib_out = nullptr;
    int tiles = 0;
    for (int i = 0; i < CV_MAXSEG; ++i) {
        ConvSeg &s = p.seg[i];
        if (i < nseg) {
            const int nb = batch ? batch[i] : 1;
            UPS_REQUIRE(x[i] && out[i] && nb > 0 && height[i] > 0 && width[i] > 0, "%s: bad feature map %d", who, i);
            s.x = x[i]; s.out = out[i]; s.res = res ? res[i] : nullptr; s.off = off ? off[i] : nullptr; s.mask = mask ? mask[i] : nullptr;
            s.w = nullptr;
            s.N = nb; s.H = height[i]; s.W = width[i];
            s.OH = s.OW = 0;
            s.Ho = (height[i] + 2 * pad - (dil * (KH - 1) + 1)) / stride + 1;
            s.Wo = (width[i] + 2 * pad - (dil * (KW - 1) + 1)) / stride + 1;
            UPS_REQUIRE(s.Ho > 0 && s.Wo > 0, "%s: empty output for feature map %d", who, i);
            UPS_REQUIRE((long)nb * height[i] * width[i] * Cin < (1L << 30), "%s: feature map %d exceeds 4 GiB (32-bit byte offsets); split the batch", who, i);
            s.M = (long)nb * s.Ho * s.Wo;
            UPS_REQUIRE(s.M * Cout < (1L << 29), "%s: output %d exceeds 2 GiB (32-bit byte offsets in the epilogue); split the batch", who, i);
            s.tile_start = tiles;
            tiles += (int)((s.M + 127) / 128);
        } else {
            s.x = s.res = s.off = s.mask = nullptr; s.w = nullptr; s.out = nullptr;
            s.N = s.H = s.W = s.Ho = s.Wo = s.OH = s.OW = 0; s.M = 0; s.tile_start = 0x7fffffff;
        }
    }
    p.m_tiles = tiles;
    return 0;
}


// out = relu?(sum_z partial[z] + bias + residual): the reduction + epilogue shared by the split-K launches (conv.hip)
int conv_splitk_reduce(hipStream_t st, const float *partial, int ksplit, long m_total, long M, int Cout, const float *bias, const float *res,
                       int relu, float *out);
