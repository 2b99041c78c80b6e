// log_table.h -- GENERATED (see the recipe at the bottom): 32 intervals of the mantissa
// m in [0.5, 1): {1/c, log(c')} with c the interval centre, 1/c rounded to double and
// c' the EXACT reciprocal of that rounded value, log(c') rounded from 60 digits.
// Used by fast_log() in fastmath.h:  m = c' (1 + r),  r = fma(m, 1/c, -1),  |r| <= 2^-6.
#pragma once
namespace covest {
__device__ const double kLogTable[64] = {
    0x1.f81f81f81f820p+0, -0x1.5af405c3649e0p-1, // m in [0.500000, 0.515625)
    0x1.e9131abf0b767p+0, -0x1.4b6fd6f970c1fp-1, // m in [0.515625, 0.531250)
    0x1.dae6076b981dbp+0, -0x1.3c6080c36bfb5p-1, // m in [0.531250, 0.546875)
    0x1.cd85689039b0bp+0, -0x1.2dbf557b0df43p-1, // m in [0.546875, 0.562500)
    0x1.c0e070381c0e0p+0, -0x1.1f8635fc61658p-1, // m in [0.562500, 0.578125)
    0x1.b4e81b4e81b4fp+0, -0x1.11af823c75aa8p-1, // m in [0.578125, 0.593750)
    0x1.a98ef606a63bep+0, -0x1.04360be7603aep-1, // m in [0.593750, 0.609375)
    0x1.9ec8e951033d9p+0, -0x1.ee2a156b413e5p-2, // m in [0.609375, 0.625000)
    0x1.948b0fcd6e9e0p+0, -0x1.d490246defa6ap-2, // m in [0.625000, 0.640625)
    0x1.8acb90f6bf3aap+0, -0x1.bb9611b80e2fcp-2, // m in [0.640625, 0.656250)
    0x1.8181818181818p+0, -0x1.a33440224fa79p-2, // m in [0.656250, 0.671875)
    0x1.78a4c8178a4c8p+0, -0x1.8b639a88b2df4p-2, // m in [0.671875, 0.687500)
    0x1.702e05c0b8170p+0, -0x1.741d876c67bb1p-2, // m in [0.687500, 0.703125)
    0x1.6816816816817p+0, -0x1.5d5bddf595f31p-2, // m in [0.703125, 0.718750)
    0x1.6058160581606p+0, -0x1.4718dc271c41cp-2, // m in [0.718750, 0.734375)
    0x1.58ed2308158edp+0, -0x1.314f1e1d35ce3p-2, // m in [0.734375, 0.750000)
    0x1.51d07eae2f815p+0, -0x1.1bf99635a6b95p-2, // m in [0.750000, 0.765625)
    0x1.4afd6a052bf5bp+0, -0x1.07138604d5864p-2, // m in [0.765625, 0.781250)
    0x1.446f86562d9fbp+0, -0x1.e530effe71013p-3, // m in [0.781250, 0.796875)
    0x1.3e22cbce4a902p+0, -0x1.bd087383bd8aap-3, // m in [0.796875, 0.812500)
    0x1.3813813813814p+0, -0x1.95a5adcf70182p-3, // m in [0.812500, 0.828125)
    0x1.323e34a2b10bfp+0, -0x1.6f0128b756ab9p-3, // m in [0.828125, 0.843750)
    0x1.2c9fb4d812ca0p+0, -0x1.4913d8333b563p-3, // m in [0.843750, 0.859375)
    0x1.27350b8812735p+0, -0x1.23d712a49c201p-3, // m in [0.859375, 0.875000)
    0x1.21fb78121fb78p+0, -0x1.fe89139dbd565p-4, // m in [0.875000, 0.890625)
    0x1.1cf06ada2811dp+0, -0x1.b6ac88dad5b1dp-4, // m in [0.890625, 0.906250)
    0x1.1811811811812p+0, -0x1.700d30aeac0e8p-4, // m in [0.906250, 0.921875)
    0x1.135c81135c811p+0, -0x1.2aa04a44717a1p-4, // m in [0.921875, 0.937500)
    0x1.0ecf56be69c90p+0, -0x1.ccb73cdddb2d0p-5, // m in [0.937500, 0.953125)
    0x1.0a6810a6810a7p+0, -0x1.466aed42de3f9p-5, // m in [0.953125, 0.968750)
    0x1.0624dd2f1a9fcp+0, -0x1.8492528c8cac5p-6, // m in [0.968750, 0.984375)
    0x1.0204081020408p+0, -0x1.010157588de69p-7, // m in [0.984375, 1.000000)
};
} // namespace covest
// recipe: for i in 0..31: c = 0.5 + (i + 0.5)/64; invc = float(1/c);
//         logc = float(ln(Fraction(1)/Fraction(invc))) with decimal precision 60.
