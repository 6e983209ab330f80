// MobileFaceNet context: state_dict ingestion (BatchNorm folding + repacking), the op schedule,
// workspace planning and the forward pass.  C-ABI entry points are at the bottom of the file.
#include <math.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace fra {

static const double kBnEps = 1e-5;  // torch BatchNorm default, face_model.py:30,42,92

enum OpKind { OP_STEM = 0, OP_DW = 1, OP_PW = 2, OP_GDCONV = 3, OP_LINEAR = 4, OP_BLOCK = 5, OP_STEMDW = 6, OP_TAIL = 7 };
// per-image activation buffers inside the workspace
enum Buf { BUF_X = -1, BUF_P0 = 0, BUF_P1 = 1, BUF_E = 2, BUF_D = 3, BUF_V = 4, BUF_OUT = 5, NBUF = 5 };
static const int64_t kBufElems[NBUF] = {56 * 56 * 64, 56 * 56 * 64, 56 * 56 * 128, 28 * 28 * 128, 512};

struct Op {
    OpKind kind;
    std::string name;  // state_dict prefix
    int in, out, resid;  // Buf ids (resid < -1: none)
    int cin, cout, hin, hout, stride;
    bool prelu;
    // device parameter pointers (into the arena)
    const void* w = nullptr;
    const float* bias = nullptr;
    const float* slope = nullptr;
    size_t w_off = 0, bias_off = 0, slope_off = 0;  // arena offsets
    double flops = 0, bytes = 0;                    // per image (activations in + out)
    double wbytes = 0;                              // parameters, read once per launch
    // OP_BLOCK only: expanded width, shortcut flag, arena offsets of the eight parameter arrays
    int cexp = 0;
    bool residual = false;
    size_t blk_off[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // w1, b1p, s1p, w2p, b2p, s2p, w3p, b3
    BlockParams bp = {};
    // OP_BLOCK run by the register-resident residual kernel (rblock): its weight packages in the arena; the packages
    // of consecutive ops of one chain are contiguous.  chain_pos = index inside the chain, chain_len = its length
    bool rblock = false;
    size_t rb_off = 0;
    const void* rb = nullptr;
    int chain_pos = 0, chain_len = 1;
    // OP_STEMDW only
    size_t sd_off[6] = {0, 0, 0, 0, 0, 0};          // wa, b1, s1, w2, b2, s2
    StemDwParams sp = {};
    // OP_TAIL only: conv_6_sep weights / bias / slope - 1, conv_6_dw weights / bias
    size_t tl_off[5] = {0, 0, 0, 0, 0};
    const void* tl[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
};

static const int NO_RESID = -100;

}  // namespace fra

struct mfn_handle {
    int device = 0;
    int E = 512;
    int dt = fra::ACT_F16;
    int chunk = 0;
    bool loaded = false;
    bool fused = true;                 // run inverted-residual blocks as one kernel each
    bool chain = true;                 // ... and runs of identical residual blocks as one launch where that pays
    std::vector<fra::Op> ops_plain;    // one kernel per conv
    std::vector<fra::Op> ops_fused;    // one kernel per block
    std::vector<fra::Op>& ops() { return fused ? ops_fused : ops_plain; }
    const std::vector<fra::Op>& ops() const { return fused ? ops_fused : ops_plain; }
    void* arena = nullptr;
    size_t arena_bytes = 0;
};

namespace fra {

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static void build_ops(mfn_handle* h, bool fused) {
    std::vector<Op>& ops = fused ? h->ops_fused : h->ops_plain;
    const int es = 2;  // bytes per activation / weight element
    auto add = [&](OpKind k, const std::string& name, int in, int out, int resid, int cin, int cout, int hin,
                   int hout, int stride, bool prelu) {
        Op o;
        o.kind = k; o.name = name; o.in = in; o.out = out; o.resid = resid;
        o.cin = cin; o.cout = cout; o.hin = hin; o.hout = hout; o.stride = stride; o.prelu = prelu;
        const double px_in = (double)hin * hin, px_out = (double)hout * hout;
        switch (k) {
            case OP_STEM:
                o.flops = 2.0 * px_out * cout * 27;
                o.bytes = px_in * 3 * 4 + px_out * cout * es;
                o.wbytes = 27 * 64 * 4;
                break;
            case OP_DW:
                o.flops = 2.0 * px_out * cout * 9;
                o.bytes = (px_in * cin + px_out * cout) * es;
                o.wbytes = 9.0 * cout * es;
                break;
            case OP_PW:
                o.flops = 2.0 * px_out * cin * cout;
                o.bytes = (px_in * cin + px_out * cout + (resid != NO_RESID ? px_out * cout : 0)) * es;
                o.wbytes = (double)cin * cout * es;
                break;
            case OP_GDCONV:
                o.flops = 2.0 * 49 * cout;
                o.bytes = (49.0 * cin + cout) * es;
                o.wbytes = 49.0 * cout * es;
                break;
            case OP_LINEAR:
                o.flops = 2.0 * cin * cout;
                o.bytes = cin * es + cout * 4.0;
                o.wbytes = (double)cin * cout * es;
                break;
            case OP_BLOCK: break;  // filled in by the caller
            case OP_STEMDW:
                o.flops = 2.0 * px_out * 64 * 27 + 2.0 * px_out * 64 * 9;
                o.bytes = px_in * 3 * 4 + px_out * 64 * es;
                o.wbytes = 64 * 32 * es + 9 * 64 * es;
                break;
            case OP_TAIL:
                o.flops = 2.0 * 49 * cin * cout + 2.0 * 49 * cout;
                o.bytes = (49.0 * cin + cout) * es;
                o.wbytes = ((double)cin * cout + 49.0 * cout) * es;
                break;
        }
        ops.push_back(o);
    };
    // face_model.py:80-81
    if (fused) {
        add(OP_STEMDW, "conv2_dw", BUF_X, BUF_P1, NO_RESID, 3, 64, 112, 56, 2, true);
    } else {
        add(OP_STEM, "conv1", BUF_X, BUF_P0, NO_RESID, 3, 64, 112, 56, 2, true);
        add(OP_DW, "conv2_dw", BUF_P0, BUF_P1, NO_RESID, 64, 64, 56, 56, 1, true);
    }
    int cur = BUF_P1;
    int chain_i = 0, chain_n = 1;   // position / length of the run of identical residual blocks being added
    auto block = [&](const std::string& p, int cin, int cexp, int cout, int stride, bool res, int hin) {
        const int hout = (hin + 2 - 3) / stride + 1;
        const int other = cur == BUF_P0 ? BUF_P1 : BUF_P0;
        if (fused) {                                                                        // face_model.py:55-65
            add(OP_BLOCK, p, cur, other, NO_RESID, cin, cout, hin, hout, stride, false);
            Op& o = ops.back();
            o.cexp = cexp;
            o.residual = res;
            const double pin = (double)hin * hin, pout = (double)hout * hout;
            o.flops = 2.0 * (pin * cin * cexp + pout * cexp * 9 + pout * cexp * cout);
            o.bytes = (pin * cin + pout * cout) * es;    // the shortcut re-reads the input tile: no extra traffic
            o.wbytes = ((double)cin * cexp + 9.0 * cexp + (double)cexp * cout) * es;
            if (res && stride == 1 && cin == 128 && (hin == 14 || hin == 7) && rblock_package_bytes(cin, cexp) > 0) {   // conv_4.*, conv_5.*
                o.rblock = true;
                o.chain_pos = chain_i;
                o.chain_len = chain_n;
            }
            cur = other;
            return;
        }
        add(OP_PW, p + ".conv", cur, BUF_E, NO_RESID, cin, cexp, hin, hin, 1, true);        // face_model.py:51
        add(OP_DW, p + ".conv_dw", BUF_E, BUF_D, NO_RESID, cexp, cexp, hin, hout, stride, true);  // :52
        add(OP_PW, p + ".project", BUF_D, other, res ? cur : NO_RESID, cexp, cout, hout, hout, 1, false);  // :53,62
        cur = other;
    };
    // face_model.py:82-87
    block("conv_23", 64, 128, 64, 2, false, 56);
    for (int i = 0; i < 4; ++i) block("conv_3.model." + std::to_string(i), 64, 128, 64, 1, true, 28);
    block("conv_34", 64, 256, 128, 2, false, 28);
    chain_n = 6;
    for (chain_i = 0; chain_i < 6; ++chain_i) block("conv_4.model." + std::to_string(chain_i), 128, 256, 128, 1, true, 14);
    chain_i = 0; chain_n = 1;
    block("conv_45", 128, 512, 128, 2, false, 14);
    chain_n = 2;
    for (chain_i = 0; chain_i < 2; ++chain_i) block("conv_5.model." + std::to_string(chain_i), 128, 256, 128, 1, true, 7);
    chain_i = 0; chain_n = 1;
    // face_model.py:88-92
    if (fused) {
        add(OP_TAIL, "conv_6_dw", cur, BUF_V, NO_RESID, 128, 512, 7, 1, 1, false);           // conv_6_sep + conv_6_dw
    } else {
        add(OP_PW, "conv_6_sep", cur, BUF_E, NO_RESID, 128, 512, 7, 7, 1, true);
        add(OP_GDCONV, "conv_6_dw", BUF_E, BUF_V, NO_RESID, 512, 512, 7, 1, 1, false);
    }
    add(OP_LINEAR, "linear", BUF_V, BUF_OUT, NO_RESID, 512, h->E, 1, 1, 1, false);
}

// ---- 16-bit conversions on the host (round to nearest even) ------------------------------------
static uint16_t f32_to_bf16_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static uint16_t f32_to_f16_bits(float f) {
    _Float16 hv = (_Float16)f;  // host compiler converts with round-to-nearest-even
    uint16_t b;
    memcpy(&b, &hv, 2);
    return b;
}
static uint16_t to16(int dt, float f) { return dt == ACT_BF16 ? f32_to_bf16_bits(f) : f32_to_f16_bits(f); }

struct TensorView {
    const float* data;
    int ndim;
    int64_t shape[4];
    int64_t numel() const {
        int64_t n = 1;
        for (int i = 0; i < ndim; ++i) n *= shape[i];
        return n;
    }
};

static int fetch(const std::map<std::string, TensorView>& m, const std::string& key, int64_t numel,
                 const float** out) {
    auto it = m.find(key);
    FRA_REQUIRE(it != m.end(), FRA_ERR_WEIGHTS, "state_dict is missing key '%s'", key.c_str());
    FRA_REQUIRE(it->second.data != nullptr, FRA_ERR_WEIGHTS, "state_dict key '%s' has a null data pointer",
                key.c_str());
    FRA_REQUIRE(it->second.numel() == numel, FRA_ERR_WEIGHTS, "state_dict key '%s' has %lld elements, expected %lld",
                key.c_str(), (long long)it->second.numel(), (long long)numel);
    *out = it->second.data;
    return FRA_OK;
}

// scale[c] = gamma / sqrt(var + eps), shift[c] = beta - mean * scale     (eval-mode BatchNorm)
static int bn_fold(const std::map<std::string, TensorView>& m, const std::string& prefix, int C,
                   std::vector<double>& scale, std::vector<double>& shift) {
    const float *g, *b, *mu, *var;
    int rc;
    if ((rc = fetch(m, prefix + ".weight", C, &g))) return rc;
    if ((rc = fetch(m, prefix + ".bias", C, &b))) return rc;
    if ((rc = fetch(m, prefix + ".running_mean", C, &mu))) return rc;
    if ((rc = fetch(m, prefix + ".running_var", C, &var))) return rc;
    scale.resize(C);
    shift.resize(C);
    for (int c = 0; c < C; ++c) {
        scale[c] = (double)g[c] / sqrt((double)var[c] + kBnEps);
        shift[c] = (double)b[c] - (double)mu[c] * scale[c];
    }
    return FRA_OK;
}

static int load_weights(mfn_handle* h, const mfn_tensor_desc* tensors, int n) {
    std::map<std::string, TensorView> m;
    for (int i = 0; i < n; ++i) {
        FRA_REQUIRE(tensors[i].name != nullptr, FRA_ERR_WEIGHTS, "tensor %d has no name", i);
        FRA_REQUIRE(tensors[i].ndim >= 0 && tensors[i].ndim <= 4, FRA_ERR_WEIGHTS, "tensor '%s': ndim %d",
                    tensors[i].name, tensors[i].ndim);
        std::string key = tensors[i].name;
        if (key.compare(0, 7, "module.") == 0) key = key.substr(7);  // nn.DataParallel checkpoints
        TensorView v;
        v.data = tensors[i].data;
        v.ndim = tensors[i].ndim;
        for (int d = 0; d < 4; ++d) v.shape[d] = d < tensors[i].ndim ? tensors[i].shape[d] : 1;
        m[key] = v;
    }
    const int dt = h->dt;
    // plan the arena (both schedules keep their own copies; 5 MB in all)
    size_t off = 0;
    std::vector<Op*> all;
    for (Op& o : h->ops_plain) all.push_back(&o);
    for (Op& o : h->ops_fused) all.push_back(&o);
    for (Op* op : all) {
        Op& o = *op;
        if (o.kind == OP_BLOCK && o.rblock) {
            // packages are whole KB, so the ops of a chain stay contiguous (the chain kernel streams across them)
            o.rb_off = off;
            off += rblock_package_bytes(o.cin, o.cexp);
            continue;
        }
        if (o.kind == OP_BLOCK) {
            const size_t sz[8] = {(size_t)o.cexp * o.cin * 2, (size_t)o.cexp * 4, (size_t)o.cexp * 4, 9 * (size_t)o.cexp * 2,
                                  (size_t)o.cexp * 4,         (size_t)o.cexp * 4, (size_t)o.cout * o.cexp * 2, (size_t)o.cout * 4};
            for (int i = 0; i < 8; ++i) { o.blk_off[i] = off; off = align_up(off + sz[i], 256); }
            continue;
        }
        if (o.kind == OP_STEMDW) {
            const size_t sz[6] = {64 * 32 * 2, 256, 256, 9 * 64 * 2, 256, 256};
            for (int i = 0; i < 6; ++i) { o.sd_off[i] = off; off = align_up(off + sz[i], 256); }
            continue;
        }
        if (o.kind == OP_TAIL) {
            const size_t sz[5] = {(size_t)o.cin * o.cout * 2, (size_t)o.cout * 4, (size_t)o.cout * 4, 49 * (size_t)o.cout * 2,
                                  (size_t)o.cout * 4};
            for (int i = 0; i < 5; ++i) { o.tl_off[i] = off; off = align_up(off + sz[i], 256); }
            continue;
        }
        size_t wbytes = 0;
        switch (o.kind) {
            case OP_STEM: wbytes = 27 * 64 * 4; break;
            case OP_DW: wbytes = 9 * (size_t)o.cout * 2; break;
            case OP_PW: wbytes = (size_t)o.cin * o.cout * 2; break;
            case OP_GDCONV: wbytes = 49 * (size_t)o.cout * 2; break;
            case OP_LINEAR: wbytes = (size_t)o.cin * o.cout * 2; break;
            case OP_BLOCK: break;
            case OP_STEMDW: break;
            case OP_TAIL: break;
        }
        o.w_off = off; off = align_up(off + wbytes, 256);
        o.bias_off = off; off = align_up(off + (size_t)o.cout * 4, 256);
        o.slope_off = off; off = align_up(off + (size_t)o.cout * 4, 256);
    }
    std::vector<unsigned char> host(off, 0);
    for (Op* op : all) {
        Op& o = *op;
        std::vector<double> scale, shift;
        const float* w = nullptr;
        int rc;
        const int C = o.cout;
        if (o.kind == OP_STEMDW) {
            const float* a;
            uint16_t* wa = reinterpret_cast<uint16_t*>(host.data() + o.sd_off[0]);
            float* b1 = reinterpret_cast<float*>(host.data() + o.sd_off[1]);
            float* s1 = reinterpret_cast<float*>(host.data() + o.sd_off[2]);
            uint16_t* w2 = reinterpret_cast<uint16_t*>(host.data() + o.sd_off[3]);
            float* b2 = reinterpret_cast<float*>(host.data() + o.sd_off[4]);
            float* s2 = reinterpret_cast<float*>(host.data() + o.sd_off[5]);
            if ((rc = fetch(m, "conv1.conv.weight", 64 * 27, &w))) return rc;
            if ((rc = bn_fold(m, "conv1.bn", 64, scale, shift))) return rc;
            if ((rc = fetch(m, "conv1.prelu.weight", 64, &a))) return rc;
            for (int oc = 0; oc < 64; ++oc) {
                // unit U = oc / 8 = 4 v + g, channel 4 a + i of it: MFMA row tile 2 v + a, row 4 g + i
                const int U = oc >> 3, row = 16 * (2 * (U >> 2) + ((oc >> 2) & 1)) + 4 * (U & 3) + (oc & 3);
                for (int t = 0; t < 32; ++t)
                    wa[row * 32 + t] = to16(dt, t < 27 ? (float)(w[oc * 27 + t] * scale[oc]) : 0.0f);
                b1[oc] = (float)shift[oc];
                s1[oc] = a[oc] - 1.0f;
            }
            if ((rc = fetch(m, "conv2_dw.conv.weight", 64 * 9, &w))) return rc;
            if ((rc = bn_fold(m, "conv2_dw.bn", 64, scale, shift))) return rc;
            if ((rc = fetch(m, "conv2_dw.prelu.weight", 64, &a))) return rc;
            for (int c = 0; c < 64; ++c) {
                for (int t = 0; t < 9; ++t) w2[t * 64 + c] = to16(dt, (float)(w[c * 9 + t] * scale[c]));
                b2[c] = (float)shift[c];
                s2[c] = a[c] - 1.0f;
            }
            continue;
        }
        if (o.kind == OP_TAIL) {
            const float* a;
            uint16_t* w6 = reinterpret_cast<uint16_t*>(host.data() + o.tl_off[0]);
            float* b6 = reinterpret_cast<float*>(host.data() + o.tl_off[1]);
            float* s6 = reinterpret_cast<float*>(host.data() + o.tl_off[2]);
            uint16_t* w7 = reinterpret_cast<uint16_t*>(host.data() + o.tl_off[3]);
            float* b7 = reinterpret_cast<float*>(host.data() + o.tl_off[4]);
            if ((rc = fetch(m, "conv_6_sep.conv.weight", (int64_t)C * o.cin, &w))) return rc;
            if ((rc = bn_fold(m, "conv_6_sep.bn", C, scale, shift))) return rc;
            if ((rc = fetch(m, "conv_6_sep.prelu.weight", C, &a))) return rc;
            for (int oc = 0; oc < C; ++oc) {
                for (int k = 0; k < o.cin; ++k) w6[(size_t)oc * o.cin + k] = to16(dt, (float)(w[(size_t)oc * o.cin + k] * scale[oc]));
                b6[oc] = (float)shift[oc];
                s6[oc] = a[oc] - 1.0f;
            }
            if ((rc = fetch(m, "conv_6_dw.conv.weight", (int64_t)C * 49, &w))) return rc;
            if ((rc = bn_fold(m, "conv_6_dw.bn", C, scale, shift))) return rc;
            for (int c = 0; c < C; ++c) {
                for (int t = 0; t < 49; ++t) w7[(size_t)t * C + c] = to16(dt, (float)(w[(size_t)c * 49 + t] * scale[c]));
                b7[c] = (float)shift[c];
            }
            continue;
        }
        if (o.kind == OP_BLOCK && o.rblock) {
            const int CE = o.cexp, CI = o.cin;
            const float *a1, *a2, *w2, *w3;
            std::vector<double> sc2, sh2, sc3, sh3;
            if ((rc = fetch(m, o.name + ".conv.conv.weight", (int64_t)CE * CI, &w))) return rc;
            if ((rc = bn_fold(m, o.name + ".conv.bn", CE, scale, shift))) return rc;
            if ((rc = fetch(m, o.name + ".conv.prelu.weight", CE, &a1))) return rc;
            if ((rc = fetch(m, o.name + ".conv_dw.conv.weight", (int64_t)CE * 9, &w2))) return rc;
            if ((rc = bn_fold(m, o.name + ".conv_dw.bn", CE, sc2, sh2))) return rc;
            if ((rc = fetch(m, o.name + ".conv_dw.prelu.weight", CE, &a2))) return rc;
            if ((rc = fetch(m, o.name + ".project.conv.weight", (int64_t)C * CE, &w3))) return rc;
            if ((rc = bn_fold(m, o.name + ".project.bn", C, sc3, sh3))) return rc;
            // the same folded values (rounded to float the same way) as the one-kernel-per-block layout below
            std::vector<float> fw1((size_t)CE * CI), fb1(CE), fw2((size_t)CE * 9), fb2(CE), fw3((size_t)C * CE), fb3(C);
            for (int e = 0; e < CE; ++e) {
                for (int k = 0; k < CI; ++k) fw1[(size_t)e * CI + k] = (float)(w[(size_t)e * CI + k] * scale[e]);
                fb1[e] = (float)shift[e];
                for (int t = 0; t < 9; ++t) fw2[(size_t)e * 9 + t] = (float)(w2[(size_t)e * 9 + t] * sc2[e]);
                fb2[e] = (float)sh2[e];
            }
            for (int n = 0; n < C; ++n) {
                for (int e = 0; e < CE; ++e) fw3[(size_t)n * CE + e] = (float)(w3[(size_t)n * CE + e] * sc3[n]);
                fb3[n] = (float)sh3[n];
            }
            if ((rc = pack_rblock(dt, CI, CE, fw1.data(), fb1.data(), a1, fw2.data(), fb2.data(), a2, fw3.data(), fb3.data(),
                                  host.data() + o.rb_off)))
                return rc;
            continue;
        }
        if (o.kind == OP_BLOCK) {
            // expand (natural row order), depthwise and project columns in the permuted channel order
            const int CE = o.cexp, CI = o.cin;
            const float* a;
            uint16_t* w1 = reinterpret_cast<uint16_t*>(host.data() + o.blk_off[0]);
            float* b1p = reinterpret_cast<float*>(host.data() + o.blk_off[1]);
            float* s1p = reinterpret_cast<float*>(host.data() + o.blk_off[2]);
            uint16_t* w2p = reinterpret_cast<uint16_t*>(host.data() + o.blk_off[3]);
            float* b2p = reinterpret_cast<float*>(host.data() + o.blk_off[4]);
            float* s2p = reinterpret_cast<float*>(host.data() + o.blk_off[5]);
            uint16_t* w3p = reinterpret_cast<uint16_t*>(host.data() + o.blk_off[6]);
            float* b3 = reinterpret_cast<float*>(host.data() + o.blk_off[7]);
            if ((rc = fetch(m, o.name + ".conv.conv.weight", (int64_t)CE * CI, &w))) return rc;
            if ((rc = bn_fold(m, o.name + ".conv.bn", CE, scale, shift))) return rc;
            if ((rc = fetch(m, o.name + ".conv.prelu.weight", CE, &a))) return rc;
            for (int e = 0; e < CE; ++e) {
                for (int k = 0; k < CI; ++k) w1[(size_t)e * CI + k] = to16(dt, (float)(w[(size_t)e * CI + k] * scale[e]));
                b1p[perm_pos(e)] = (float)shift[e];
                s1p[perm_pos(e)] = a[e] - 1.0f;   // kernels compute v + (slope-1) min(v,0)
            }
            if ((rc = fetch(m, o.name + ".conv_dw.conv.weight", (int64_t)CE * 9, &w))) return rc;
            if ((rc = bn_fold(m, o.name + ".conv_dw.bn", CE, scale, shift))) return rc;
            if ((rc = fetch(m, o.name + ".conv_dw.prelu.weight", CE, &a))) return rc;
            for (int e = 0; e < CE; ++e) {
                for (int t = 0; t < 9; ++t) w2p[(size_t)t * CE + perm_pos(e)] = to16(dt, (float)(w[(size_t)e * 9 + t] * scale[e]));
                b2p[perm_pos(e)] = (float)shift[e];
                s2p[perm_pos(e)] = a[e] - 1.0f;
            }
            if ((rc = fetch(m, o.name + ".project.conv.weight", (int64_t)C * CE, &w))) return rc;
            if ((rc = bn_fold(m, o.name + ".project.bn", C, scale, shift))) return rc;
            for (int n = 0; n < C; ++n) {
                for (int e = 0; e < CE; ++e) w3p[w3_image_index(n, perm_pos(e), C)] = to16(dt, (float)(w[(size_t)n * CE + e] * scale[n]));
                b3[n] = (float)shift[n];
            }
            continue;
        }
        float* bias = reinterpret_cast<float*>(host.data() + o.bias_off);
        float* slope = reinterpret_cast<float*>(host.data() + o.slope_off);
        if (o.kind == OP_LINEAR) {
            if ((rc = fetch(m, "linear.weight", (int64_t)o.cin * C, &w))) return rc;
            if ((rc = bn_fold(m, "bn", C, scale, shift))) return rc;
            uint16_t* dst = reinterpret_cast<uint16_t*>(host.data() + o.w_off);
            for (int e = 0; e < C; ++e)
                for (int k = 0; k < o.cin; ++k) dst[(size_t)e * o.cin + k] = to16(dt, (float)(w[(size_t)e * o.cin + k] * scale[e]));
            for (int e = 0; e < C; ++e) bias[e] = (float)shift[e];
            continue;
        }
        const int kk = o.kind == OP_STEM ? 27 : o.kind == OP_DW ? 9 : o.kind == OP_GDCONV ? 49 : o.cin;
        if ((rc = fetch(m, o.name + ".conv.weight", (int64_t)C * kk, &w))) return rc;
        if ((rc = bn_fold(m, o.name + ".bn", C, scale, shift))) return rc;
        for (int c = 0; c < C; ++c) bias[c] = (float)shift[c];
        if (o.prelu) {
            const float* a;
            if ((rc = fetch(m, o.name + ".prelu.weight", C, &a))) return rc;
            for (int c = 0; c < C; ++c) slope[c] = a[c];
        }
        if (o.kind == OP_STEM) {
            // torch [oc][c][ky][kx] -> [tap = (c*3+ky)*3+kx][oc], fp32
            float* dst = reinterpret_cast<float*>(host.data() + o.w_off);
            for (int oc = 0; oc < 64; ++oc)
                for (int t = 0; t < 27; ++t) dst[t * 64 + oc] = (float)(w[oc * 27 + t] * scale[oc]);
        } else if (o.kind == OP_DW || o.kind == OP_GDCONV) {
            // torch [c][1][ky][kx] -> [tap][c]
            uint16_t* dst = reinterpret_cast<uint16_t*>(host.data() + o.w_off);
            for (int c = 0; c < C; ++c)
                for (int t = 0; t < kk; ++t) dst[(size_t)t * C + c] = to16(dt, (float)(w[(size_t)c * kk + t] * scale[c]));
        } else {  // OP_PW: torch [oc][ic][1][1] keeps its layout, rows scaled
            uint16_t* dst = reinterpret_cast<uint16_t*>(host.data() + o.w_off);
            for (int oc = 0; oc < C; ++oc)
                for (int k = 0; k < o.cin; ++k) dst[(size_t)oc * o.cin + k] = to16(dt, (float)(w[(size_t)oc * o.cin + k] * scale[oc]));
        }
    }
    FRA_DEVICE(h->device);
    // a reload overwrites the arena forwards may still be reading (any stream): let the device drain first
    if (h->loaded) FRA_HIP(hipDeviceSynchronize());
    if (h->arena_bytes < off) {
        if (h->arena) FRA_HIP(fra::dev_free(h->arena));
        h->arena = nullptr;
        FRA_HIP(fra::dev_malloc(&h->arena, off));
        h->arena_bytes = off;
    }
    FRA_HIP(hipMemcpy(h->arena, host.data(), off, hipMemcpyHostToDevice));
    for (Op* op : all) {
        Op& o = *op;
        char* base = static_cast<char*>(h->arena);
        if (o.kind == OP_STEMDW) {
            o.sp.wa = base + o.sd_off[0];
            o.sp.b1 = reinterpret_cast<const float*>(base + o.sd_off[1]);
            o.sp.s1 = reinterpret_cast<const float*>(base + o.sd_off[2]);
            o.sp.w2 = base + o.sd_off[3];
            o.sp.b2 = reinterpret_cast<const float*>(base + o.sd_off[4]);
            o.sp.s2 = reinterpret_cast<const float*>(base + o.sd_off[5]);
            continue;
        }
        if (o.kind == OP_TAIL) {
            for (int i = 0; i < 5; ++i) o.tl[i] = base + o.tl_off[i];
            continue;
        }
        if (o.kind == OP_BLOCK && o.rblock) {
            o.rb = base + o.rb_off;
            continue;
        }
        if (o.kind == OP_BLOCK) {
            BlockParams& b = o.bp;
            b.cin = o.cin; b.cexp = o.cexp; b.cout = o.cout; b.h = o.hin; b.stride = o.stride; b.residual = o.residual;
            b.w1 = base + o.blk_off[0];
            b.b1p = reinterpret_cast<const float*>(base + o.blk_off[1]);
            b.s1p = reinterpret_cast<const float*>(base + o.blk_off[2]);
            b.w2p = base + o.blk_off[3];
            b.b2p = reinterpret_cast<const float*>(base + o.blk_off[4]);
            b.s2p = reinterpret_cast<const float*>(base + o.blk_off[5]);
            b.w3p = base + o.blk_off[6];
            b.b3 = reinterpret_cast<const float*>(base + o.blk_off[7]);
            continue;
        }
        o.w = static_cast<char*>(h->arena) + o.w_off;
        o.bias = reinterpret_cast<const float*>(static_cast<char*>(h->arena) + o.bias_off);
        o.slope = o.prelu ? reinterpret_cast<const float*>(static_cast<char*>(h->arena) + o.slope_off) : nullptr;
    }
    h->loaded = true;
    return FRA_OK;
}

// workspace layout for a slice of `b` images: the five buffers back to back, 256-byte aligned
struct WsPlan {
    size_t off[NBUF];
    size_t emb;      // f32 [b][E] pre-combine embeddings (flip-TTA only)
    size_t total;
};
static WsPlan plan_ws(int b, int E, bool tta) {
    WsPlan p;
    size_t off = 0;
    for (int i = 0; i < NBUF; ++i) {
        p.off[i] = off;
        off = align_up(off + (size_t)kBufElems[i] * 2 * (size_t)b, 256);
    }
    p.emb = off;
    if (tta) off = align_up(off + (size_t)b * E * 4, 256);
    p.total = off;
    return p;
}

// Which activation buffers a launch reads and writes.  The ops of the un-fused schedule name theirs; the fused blocks
// alternate between BUF_P0 and BUF_P1 per LAUNCH (a launch may cover several blocks), starting from where the stem wrote.
struct ActCursor {
    int cur = -1;
    void next(const Op& o, int* in_id, int* out_id) {
        *in_id = o.in;
        *out_id = o.out;
        if (o.kind == OP_BLOCK && cur >= 0) {
            *in_id = cur;
            *out_id = cur == BUF_P0 ? BUF_P1 : BUF_P0;
        } else if (o.kind == OP_TAIL && cur >= 0) {
            *in_id = cur;
        }
        if (o.kind == OP_STEMDW || (o.kind == OP_BLOCK && cur >= 0)) cur = *out_id;
    }
};

static int slice_size(const mfn_handle* h, int B) { return (h->chunk > 0 && h->chunk < B) ? h->chunk : B; }

// b = images in the activation buffers; x holds n_src source crops (b == n_src, or 2 n_src with mirrored copies)
// Blocks that op i's launch covers in a batch of b images when at most `limit` ops may run: a run of register-resident
// residual blocks goes out as one launch where whole-image workgroups pay.
static int ops_in_launch(const mfn_handle* h, int i, int b, int limit) {
    const Op& o = h->ops()[i];
    if (o.kind != OP_BLOCK || !o.rblock || o.chain_len - o.chain_pos < 2 || !h->chain) return 1;
    const int left = o.chain_len - o.chain_pos;
    return rblock_chain_blocks(o.cin, o.cexp, o.hin, b, left < limit ? left : limit);
}

// `nblk` > 1: op o and the nblk - 1 ops after it as one launch, written where the last of them writes
// in_id / out_id >= 0: the activation buffers a fused block (or the tail) reads and writes, in place of the op's own --
// a launch that covers several blocks moves the activation once, so the two buffers alternate per LAUNCH
static int run_op(const mfn_handle* h, const Op& o, const void* x, int x_is_u8, int n_src, float* out, char* ws,
                  const WsPlan& p, int b, hipStream_t s, int nblk = 1, int in_id = -1, int out_id = -1) {
    auto buf = [&](int id) -> void* { return ws + p.off[id]; };
    if (in_id < 0) in_id = o.in;
    if (out_id < 0) out_id = o.out;
    switch (o.kind) {
        case OP_STEM:
            FRA_REQUIRE(!x_is_u8 && b == n_src, FRA_ERR_STATE, "uint8 input and fused flip-TTA need the fused schedule");
            return launch_stem(h->dt, (const float*)x, (const float*)o.w, o.bias, o.slope, buf(o.out), b, s);
        case OP_DW:
            return launch_dw3x3(h->dt, buf(o.in), o.w, o.bias, o.slope, buf(o.out), b, o.hin, o.hin, o.cin, o.stride, s);
        case OP_PW:
            return launch_pw(h->dt, buf(o.in), o.w, o.bias, o.slope, o.resid != NO_RESID ? buf(o.resid) : nullptr,
                             buf(o.out), (int64_t)b * o.hout * o.hout, o.cin, o.cout, s);
        case OP_GDCONV:
            return launch_gdconv7(h->dt, buf(o.in), o.w, o.bias, buf(o.out), b, o.cout, s);
        case OP_LINEAR:
            return launch_linear_l2norm(h->dt, buf(o.in), o.w, o.bias, out, b, o.cin, o.cout, s);
        case OP_BLOCK:
            if (o.rblock) return launch_rblock(h->dt, o.cin, o.cexp, o.hin, buf(in_id), o.rb, nblk, buf(out_id), b, s);
            return launch_block(h->dt, o.bp, buf(in_id), buf(out_id), b, s);
        case OP_STEMDW:
            return launch_stemdw(h->dt, o.sp, x, x_is_u8, n_src, buf(o.out), b, s);
        case OP_TAIL:
            return launch_tail(h->dt, buf(in_id), o.tl[0], (const float*)o.tl[1], (const float*)o.tl[2], o.tl[3],
                               (const float*)o.tl[4], buf(o.out), b, s);
    }
    return FRA_ERR_ARG;
}

static int check_forward_args(const mfn_handle* h, const void* x, int B, bool tta, const void* out, const void* ws,
                              size_t ws_bytes) {
    FRA_REQUIRE(h != nullptr, FRA_ERR_ARG, "null handle");
    FRA_REQUIRE(h->loaded, FRA_ERR_STATE, "mfn_forward called before mfn_load_weights");
    FRA_REQUIRE(x && out && ws, FRA_ERR_ARG, "mfn_forward: null pointer");
    FRA_REQUIRE(B > 0, FRA_ERR_ARG, "mfn_forward: batch %d", B);
    const int sl0 = slice_size(h, B);
    const size_t need = plan_ws(tta ? 2 * sl0 : sl0, h->E, tta).total;
    FRA_REQUIRE(ws_bytes >= need, FRA_ERR_WORKSPACE, "workspace has %zu bytes, batch %d needs %zu", ws_bytes, B, need);
    FRA_REQUIRE(((uintptr_t)ws & 255) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0, FRA_ERR_ARG,
                "mfn_forward: workspace must be 256-byte aligned, input/output 16-byte aligned");
    return FRA_OK;
}

static int forward(mfn_handle* h, const void* x, int x_is_u8, int B, bool tta, float* out, void* ws, size_t ws_bytes,
                   hipStream_t s, float* op_ms, int n_ops_timed) {
    int rc = check_forward_args(h, x, B, tta, out, ws, ws_bytes);
    if (rc) return rc;
    if ((rc = check_device_fault(h->device))) return rc;   // an earlier launch's wait gave up: do not pile results on top
    FRA_DEVICE(h->device);   // launches, LDS limits and occupancy answers all belong to the handle's device
    FRA_REQUIRE(h->fused || (!tta && !x_is_u8), FRA_ERR_STATE, "uint8 input and fused flip-TTA need the fused schedule");
    const int nops = (int)h->ops().size();
    const int sl = slice_size(h, B);
    struct Events {                                    // destroyed on every way out, error returns included
        std::vector<hipEvent_t> v;
        ~Events() { for (auto e : v) if (e) (void)hipEventDestroy(e); }
    } evs;
    std::vector<hipEvent_t>& ev = evs.v;
    if (op_ms) {
        FRA_REQUIRE(n_ops_timed >= nops, FRA_ERR_ARG, "op_ms has room for %d ops, model has %d", n_ops_timed, nops);
        for (int i = 0; i < nops; ++i) op_ms[i] = 0.0f;
        ev.assign(nops + 1, nullptr);
        for (auto& e : ev) FRA_HIP(hipEventCreate(&e));
    }
    for (int b0 = 0; b0 < B; b0 += sl) {
        const int b = (B - b0) < sl ? (B - b0) : sl;      // source crops in this slice
        const int nb = tta ? 2 * b : b;                   // images in flight (second half: mirrored copies)
        const WsPlan p = plan_ws(nb, h->E, tta);
        const void* xs = x_is_u8 ? static_cast<const void*>(static_cast<const unsigned char*>(x) + (size_t)b0 * 3 * 112 * 112)
                                 : static_cast<const void*>(static_cast<const float*>(x) + (size_t)b0 * 3 * 112 * 112);
        float* os = out + (size_t)b0 * h->E;
        float* emb = tta ? reinterpret_cast<float*>(static_cast<char*>(ws) + p.emb) : os;
        if (op_ms) FRA_HIP(hipEventRecord(ev[0], s));
        ActCursor act;
        for (int i = 0; i < nops;) {
            const int n = ops_in_launch(h, i, nb, nops - i);
            int in_id, out_id;
            act.next(h->ops()[i], &in_id, &out_id);
            rc = run_op(h, h->ops()[i], xs, x_is_u8, b, emb, static_cast<char*>(ws), p, nb, s, n, in_id, out_id);
            if (rc) return rc;
            if (i + n == nops && tta) {   // l2_norm(emb + emb_mirror): facebank.py:62
                rc = launch_add_l2norm(emb, emb + (size_t)b * h->E, os, b, h->E, s);
                if (rc) return rc;
            }
            // a launch that covers several ops is booked on its first op; the others read 0 ms
            if (op_ms) for (int j = 1; j <= n; ++j) FRA_HIP(hipEventRecord(ev[i + j], s));
            i += n;
        }
        if (op_ms) {
            FRA_HIP(hipEventSynchronize(ev[nops]));
            for (int i = 0; i < nops; ++i) {
                float ms = 0;
                FRA_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
                op_ms[i] += ms;
            }
        }
    }
    return FRA_OK;
}

}  // namespace fra

using namespace fra;

extern "C" {

int mfn_create(mfn_handle** out, int device, int embedding_size, int act_dtype) {
    FRA_REQUIRE(out != nullptr, FRA_ERR_ARG, "mfn_create: null output pointer");
    *out = nullptr;
    FRA_REQUIRE(act_dtype == ACT_F16 || act_dtype == ACT_BF16, FRA_ERR_ARG,
                "mfn_create: act_dtype %d unsupported (1 = bf16, 2 = f16)", act_dtype);
    FRA_REQUIRE(embedding_size == 128 || embedding_size == 256 || embedding_size == 512 || embedding_size == 1024,
                FRA_ERR_ARG, "mfn_create: embedding_size %d unsupported (128, 256, 512, 1024)", embedding_size);
    int count = 0;
    FRA_HIP(hipGetDeviceCount(&count));
    FRA_REQUIRE(device >= 0 && device < count, FRA_ERR_ARG, "mfn_create: device %d not in [0,%d)", device, count);
    mfn_handle* h = new mfn_handle();
    h->device = device;
    h->E = embedding_size;
    h->dt = act_dtype;
    build_ops(h, false);
    build_ops(h, true);
    *out = h;
    return FRA_OK;
}

int mfn_load_weights(mfn_handle* h, const mfn_tensor_desc* tensors, int n) {
    FRA_REQUIRE(h && tensors && n > 0, FRA_ERR_ARG, "mfn_load_weights: bad argument");
    return load_weights(h, tensors, n);
}

int mfn_workspace_bytes(const mfn_handle* h, int max_batch, size_t* bytes) {
    FRA_REQUIRE(h && bytes && max_batch > 0, FRA_ERR_ARG, "mfn_workspace_bytes: bad argument");
    *bytes = plan_ws(slice_size(h, max_batch), h->E, false).total;
    return FRA_OK;
}

int mfn_workspace_bytes_ex(const mfn_handle* h, int max_batch, int tta, size_t* bytes) {
    FRA_REQUIRE(h && bytes && max_batch > 0, FRA_ERR_ARG, "mfn_workspace_bytes_ex: bad argument");
    const int sl = slice_size(h, max_batch);
    *bytes = plan_ws(tta ? 2 * sl : sl, h->E, tta != 0).total;
    return FRA_OK;
}

int mfn_set_batch_chunk(mfn_handle* h, int images) {
    FRA_REQUIRE(h && images >= 0, FRA_ERR_ARG, "mfn_set_batch_chunk: bad argument");
    h->chunk = images;
    return FRA_OK;
}

int mfn_set_fused(mfn_handle* h, int fused) {
    FRA_REQUIRE(h != nullptr, FRA_ERR_ARG, "mfn_set_fused: null handle");
    h->fused = fused != 0;
    return FRA_OK;
}

int mfn_set_chain(mfn_handle* h, int chain) {
    FRA_REQUIRE(h != nullptr, FRA_ERR_ARG, "mfn_set_chain: null handle");
    h->chain = chain != 0;
    return FRA_OK;
}

int mfn_launch_plan(const mfn_handle* h, int B, int* first_op, int n_ops) {
    FRA_REQUIRE(h && first_op && B > 0, FRA_ERR_ARG, "mfn_launch_plan: bad argument");
    const int nops = (int)h->ops().size();
    FRA_REQUIRE(n_ops >= nops, FRA_ERR_ARG, "mfn_launch_plan: room for %d ops, model has %d", n_ops, nops);
    for (int i = 0; i < nops;) {
        const int n = ops_in_launch(h, i, B, nops - i);
        for (int j = 0; j < n; ++j) first_op[i + j] = i;
        i += n;
    }
    return FRA_OK;
}

int mfn_forward(mfn_handle* h, const float* x, int B, float* out, void* ws, size_t ws_bytes, fra_stream stream) {
    return forward(h, x, 0, B, false, out, ws, ws_bytes, (hipStream_t)stream, nullptr, 0);
}

int mfn_forward_ex(mfn_handle* h, const void* x, int input_format, int B, int tta, float* out, void* ws,
                   size_t ws_bytes, fra_stream stream) {
    FRA_REQUIRE(input_format == 0 || input_format == 1, FRA_ERR_ARG, "mfn_forward_ex: input_format %d (0 = f32 NCHW, 1 = u8 HWC)",
                input_format);
    return forward(h, x, input_format, B, tta != 0, out, ws, ws_bytes, (hipStream_t)stream, nullptr, 0);
}

int mfn_forward_timed(mfn_handle* h, const float* x, int B, float* out, void* ws, size_t ws_bytes, fra_stream stream,
                      float* op_ms, int n_ops) {
    FRA_REQUIRE(op_ms != nullptr, FRA_ERR_ARG, "mfn_forward_timed: null op_ms");
    return forward(h, x, 0, B, false, out, ws, ws_bytes, (hipStream_t)stream, op_ms, n_ops);
}

int mfn_op_count(const mfn_handle* h) { return h ? (int)h->ops().size() : -1; }

int mfn_op_variant(const mfn_handle* h, int op, int B) {
    FRA_REQUIRE(h && op >= 0 && op < (int)h->ops().size() && B > 0, FRA_ERR_ARG, "mfn_op_variant: bad argument");
    const Op& o = h->ops()[op];
    if (o.kind != OP_BLOCK) return 0;
    if (o.rblock) return 1;
    FRA_DEVICE(h->device);
    return block_uses_pipeline(o.bp, B) ? 2 : 0;
}

int mfn_op_info(const mfn_handle* h, int op, char* name, int name_cap, int* kind, double* flops, double* bytes,
                double* weight_bytes) {
    FRA_REQUIRE(h && op >= 0 && op < (int)h->ops().size(), FRA_ERR_ARG, "mfn_op_info: bad argument");
    const Op& o = h->ops()[op];
    if (name && name_cap > 0) {
        strncpy(name, o.name.c_str(), name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (kind) *kind = (int)o.kind;
    if (flops) *flops = o.flops;
    if (bytes) *bytes = o.bytes;
    if (weight_bytes) *weight_bytes = o.wbytes;
    return FRA_OK;
}

int mfn_debug_prefix(mfn_handle* h, const float* x, int B, int n_ops, float* out_f32, int64_t* numel_per_image,
                     void* ws, size_t ws_bytes, fra_stream stream) {
    int rc = check_forward_args(h, x, B, false, out_f32, ws, ws_bytes);
    if (rc) return rc;
    FRA_REQUIRE(n_ops >= 1 && n_ops < (int)h->ops().size(), FRA_ERR_ARG,
                "mfn_debug_prefix: n_ops %d not in [1,%d)", n_ops, (int)h->ops().size());
    FRA_REQUIRE(slice_size(h, B) == B, FRA_ERR_STATE, "mfn_debug_prefix: batch chunking must be off");
    FRA_DEVICE(h->device);
    const WsPlan p = plan_ws(B, h->E, false);
    hipStream_t s = (hipStream_t)stream;
    ActCursor act;
    int out_id = -1;
    for (int i = 0; i < n_ops;) {
        const int n = ops_in_launch(h, i, B, n_ops - i);
        int in_id;
        act.next(h->ops()[i], &in_id, &out_id);
        rc = run_op(h, h->ops()[i], x, 0, B, nullptr, static_cast<char*>(ws), p, B, s, n, in_id, out_id);
        if (rc) return rc;
        i += n;
    }
    const Op& o = h->ops()[n_ops - 1];
    const int64_t per = (int64_t)o.hout * o.hout * o.cout;
    if (numel_per_image) *numel_per_image = per;
    return launch_to_f32(h->dt, static_cast<char*>(ws) + p.off[out_id], out_f32, per * B, s);
}

int mfn_destroy(mfn_handle* h) {
    if (!h) return FRA_OK;
    const int device = h->device;
    if (h->arena) {
        fra::DeviceGuard guard(h->device);
        (void)fra::dev_free(h->arena);           // waits for the device to drain: every launch of this handle has finished
    }
    delete h;
    return check_device_fault(device);
}

int mfn_tta_combine(const float* a, const float* b, int B, int E, float* out, fra_stream stream) {
    return launch_add_l2norm(a, b, out, B, E, (hipStream_t)stream);
}

int mfn_op_stem(int dtype, const float* x, const float* w, const float* bias, const float* slope, void* out, int B,
                fra_stream stream) {
    return launch_stem(dtype, x, w, bias, slope, out, B, (hipStream_t)stream);
}
int mfn_op_dw3x3(int dtype, const void* in, const void* w, const float* bias, const float* slope, void* out, int B,
                 int H, int W, int C, int stride, fra_stream stream) {
    return launch_dw3x3(dtype, in, w, bias, slope, out, B, H, W, C, stride, (hipStream_t)stream);
}
int mfn_op_pw(int dtype, const void* in, const void* w, const float* bias, const float* slope, const void* resid,
              void* out, int64_t M, int K, int N, fra_stream stream) {
    return launch_pw(dtype, in, w, bias, slope, resid, out, M, K, N, (hipStream_t)stream);
}
int mfn_op_block(int dtype, int cin, int cexp, int cout, int h, int stride, int residual, const void* x,
                 const void* w1, const float* b1p, const float* s1p, const void* w2p, const float* b2p,
                 const float* s2p, const void* w3p, const float* b3, void* out, int B, fra_stream stream) {
    BlockParams p;
    p.cin = cin; p.cexp = cexp; p.cout = cout; p.h = h; p.stride = stride; p.residual = residual;
    p.w1 = w1; p.b1p = b1p; p.s1p = s1p; p.w2p = w2p; p.b2p = b2p; p.s2p = s2p; p.w3p = w3p; p.b3 = b3;
    return launch_block(dtype, p, x, out, B, (hipStream_t)stream);
}
int64_t mfn_rblock_package_bytes(int c, int cexp) { return (int64_t)rblock_package_bytes(c, cexp); }
int mfn_pack_rblock(int dtype, int c, int cexp, const float* w1, const float* b1, const float* s1, const float* w2,
                    const float* b2, const float* s2, const float* w3, const float* b3, void* packages_host) {
    return pack_rblock(dtype, c, cexp, w1, b1, s1, w2, b2, s2, w3, b3, packages_host);
}
int mfn_op_rblock(int dtype, int c, int cexp, int h, const void* x, const void* packages, int nblk, void* out, int B,
                  fra_stream stream) {
    return launch_rblock(dtype, c, cexp, h, x, packages, nblk, out, B, (hipStream_t)stream);
}
int mfn_op_wblock(int dtype, int c, int cexp, int h, const void* x, const void* packages, int nblk, void* out, int B,
                  fra_stream stream) {
    return launch_wblock(dtype, c, cexp, h, x, packages, nblk, out, B, (hipStream_t)stream);
}
int mfn_op_gdconv7(int dtype, const void* in, const void* w, const float* bias, void* out, int B, int C,
                   fra_stream stream) {
    return launch_gdconv7(dtype, in, w, bias, out, B, C, (hipStream_t)stream);
}
int mfn_op_tail(int dtype, const void* x, const void* w6, const float* b6, const float* s6, const void* w7,
                const float* b7, void* out, int B, fra_stream stream) {
    return launch_tail(dtype, x, w6, b6, s6, w7, b7, out, B, (hipStream_t)stream);
}
int mfn_op_linear_l2norm(int dtype, const void* in, const void* w, const float* bias, float* out, int B, int K, int E,
                         fra_stream stream) {
    return launch_linear_l2norm(dtype, in, w, bias, out, B, K, E, (hipStream_t)stream);
}

}  // extern "C"
