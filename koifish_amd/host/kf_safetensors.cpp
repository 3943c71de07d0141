// kf_safetensors.cpp -- see kf_safetensors.hpp.  Plain C++17 (mmap + a small JSON reader); device work goes through the C ABI.
#include "kf_safetensors.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "kf_host.hpp"

namespace koifish {

// ------------------------------------------------------------------------------------------------ JSON
namespace {
struct Parser {
    const char* p;
    const char* end;
    const char* base;
    std::string err;
    bool fail(const char* what) {
        char buf[96];
        snprintf(buf, sizeof(buf), "%s at byte %zu", what, (size_t)(p - base));
        err = buf;
        return false;
    }
    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++;
    }
    bool str(std::string& out) {
        if (p >= end || *p != '"') return fail("expected string");
        p++;
        out.clear();
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("unterminated escape");
                switch (*p) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': { /* \uXXXX: keep ASCII, replace the rest (tensor names and dtypes are ASCII) */
                        if (end - p < 5) return fail("short \\u escape");
                        unsigned v = 0;
                        for (int i = 1; i <= 4; i++) {
                            const char c = p[i];
                            v = v * 16 + (c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : 99);
                        }
                        out += v < 128 ? (char)v : '?';
                        p += 4;
                        break;
                    }
                    default: out += *p;
                }
                p++;
            } else {
                out += *p++;
            }
        }
        if (p >= end) return fail("unterminated string");
        p++;
        return true;
    }
    bool value(JSON& v, int depth) {
        if (depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        if (*p == '{') {
            v.kind = JSON::OBJ;
            p++;
            ws();
            if (p < end && *p == '}') {
                p++;
                return true;
            }
            for (;;) {
                ws();
                std::string k;
                if (!str(k)) return false;
                ws();
                if (p >= end || *p != ':') return fail("expected ':'");
                p++;
                JSON child;
                if (!value(child, depth + 1)) return false;
                v.obj.emplace_back(std::move(k), std::move(child));
                ws();
                if (p < end && *p == ',') {
                    p++;
                    continue;
                }
                if (p < end && *p == '}') {
                    p++;
                    return true;
                }
                return fail("expected ',' or '}'");
            }
        }
        if (*p == '[') {
            v.kind = JSON::ARR;
            p++;
            ws();
            if (p < end && *p == ']') {
                p++;
                return true;
            }
            for (;;) {
                JSON child;
                if (!value(child, depth + 1)) return false;
                v.arr.push_back(std::move(child));
                ws();
                if (p < end && *p == ',') {
                    p++;
                    continue;
                }
                if (p < end && *p == ']') {
                    p++;
                    return true;
                }
                return fail("expected ',' or ']'");
            }
        }
        if (*p == '"') {
            v.kind = JSON::STR;
            return str(v.str);
        }
        if (end - p >= 4 && !strncmp(p, "true", 4)) {
            v.kind = JSON::BOOL, v.b = true, p += 4;
            return true;
        }
        if (end - p >= 5 && !strncmp(p, "false", 5)) {
            v.kind = JSON::BOOL, v.b = false, p += 5;
            return true;
        }
        if (end - p >= 4 && !strncmp(p, "null", 4)) {
            v.kind = JSON::NUL, p += 4;
            return true;
        }
        if (*p == '-' || (*p >= '0' && *p <= '9')) {
            char buf[64];
            size_t n = 0;
            while (p < end && n < sizeof(buf) - 1 && (*p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E' || (*p >= '0' && *p <= '9'))) buf[n++] = *p++;
            buf[n] = 0;
            char* e = nullptr;
            v.kind = JSON::NUM, v.num = strtod(buf, &e);
            if (e == buf) return fail("bad number");
            return true;
        }
        return fail("unexpected character");
    }
};
}  // namespace

bool JSON::Parse(const char* text, size_t n, JSON& out, std::string& err) {
    Parser ps{text, text + n, text, {}};
    out = JSON();
    if (!ps.value(out, 0)) {
        err = ps.err;
        return false;
    }
    ps.ws();
    if (ps.p != ps.end) {
        ps.fail("trailing characters");
        err = ps.err;
        return false;
    }
    return true;
}
const JSON* JSON::get(const std::string& key) const {
    if (kind != OBJ) return nullptr;
    for (const auto& kv : obj)
        if (kv.first == key) return &kv.second;
    return nullptr;
}
double JSON::number_or(const std::string& key, double dflt) const {
    const JSON* v = get(key);
    return v && v->kind == NUM ? v->num : dflt;
}
bool JSON::bool_or(const std::string& key, bool dflt) const {
    const JSON* v = get(key);
    return v && v->kind == BOOL ? v->b : dflt;
}

// ------------------------------------------------------------------------------------------------ safetensors
K_SafeTensors::~K_SafeTensors() {
    for (auto& f : files) {
        if (f.map && f.map != MAP_FAILED) munmap(f.map, f.size);
        if (f.fd >= 0) close(f.fd);
    }
}

static size_t dtype_bytes(const std::string& d) {
    if (d == "BF16" || d == "F16" || d == "I16" || d == "U16") return 2;
    if (d == "F32" || d == "I32" || d == "U32") return 4;
    if (d == "F64" || d == "I64" || d == "U64") return 8;
    if (d == "I8" || d == "U8" || d == "BOOL" || d == "F8_E5M2" || d == "F8_E4M3") return 1;
    return 0;
}

int K_SafeTensors::OpenFile(const std::string& path) {
    File f;
    f.path = path;
    f.fd = open(path.c_str(), O_RDONLY);
    if (f.fd < 0) {
        err = "cannot open " + path;
        return KF_INVALID_ARGS;
    }
    struct stat sb;
    if (fstat(f.fd, &sb) != 0 || sb.st_size < 8) {
        close(f.fd);
        err = path + ": shorter than a safetensors header";
        return KF_INVALID_ARGS;
    }
    f.size = (size_t)sb.st_size;
    f.map = mmap(nullptr, f.size, PROT_READ, MAP_PRIVATE, f.fd, 0);
    if (f.map == MAP_FAILED) {
        close(f.fd);
        err = "mmap failed for " + path;
        return KF_INTERNAL_ERR;
    }
    const unsigned char* b = reinterpret_cast<const unsigned char*>(f.map);
    uint64_t hlen = 0;
    for (int i = 7; i >= 0; i--) hlen = (hlen << 8) | b[i];
    if (hlen == 0 || hlen > f.size - 8) {
        munmap(f.map, f.size), close(f.fd);
        err = path + ": header length exceeds the file";
        return KF_INVALID_ARGS;
    }
    f.data_base = 8 + (size_t)hlen;
    JSON hdr;
    std::string jerr;
    if (!JSON::Parse(reinterpret_cast<const char*>(b + 8), (size_t)hlen, hdr, jerr) || hdr.kind != JSON::OBJ) {
        munmap(f.map, f.size), close(f.fd);
        err = path + ": bad JSON header (" + jerr + ")";
        return KF_INVALID_ARGS;
    }
    const int file_id = (int)files.size();
    const size_t avail = f.size - f.data_base;
    for (const auto& kv : hdr.obj) {
        if (kv.first == "__metadata__") {
            for (const auto& m : kv.second.obj)
                if (m.second.kind == JSON::STR) metadata[m.first] = m.second.str;
            continue;
        }
        const JSON *dt = kv.second.get("dtype"), *sh = kv.second.get("shape"), *off = kv.second.get("data_offsets");
        if (!dt || dt->kind != JSON::STR || !sh || sh->kind != JSON::ARR || !off || off->kind != JSON::ARR || off->arr.size() != 2) {
            munmap(f.map, f.size), close(f.fd);
            err = path + ": tensor entry '" + kv.first + "' lacks dtype/shape/data_offsets";
            return KF_INVALID_ARGS;
        }
        ST_Tensor t;
        t.name = kv.first, t.dtype = dt->str, t.file = file_id;
        size_t count = 1;
        for (const auto& d : sh->arr) {
            t.shape.push_back((int64_t)d.num);
            count *= (size_t)d.num;
        }
        t.begin = (size_t)off->arr[0].num, t.end = (size_t)off->arr[1].num;
        const size_t eb = dtype_bytes(t.dtype);
        if (t.begin > t.end || t.end > avail || (eb && t.end - t.begin != count * eb)) {
            munmap(f.map, f.size), close(f.fd);
            err = path + ": data_offsets of '" + kv.first + "' do not fit its shape or the file";
            return KF_INVALID_ARGS;
        }
        index[t.name] = (int)tensors.size();
        tensors.push_back(std::move(t));
    }
    files.push_back(f);
    return KF_OK;
}

static bool file_exists(const std::string& p) {
    struct stat sb;
    return stat(p.c_str(), &sb) == 0;
}
static bool read_text(const std::string& path, std::string& out) {
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) return false;
    char buf[65536];
    size_t n;
    out.clear();
    while ((n = fread(buf, 1, sizeof(buf), fp)) > 0) out.append(buf, n);
    fclose(fp);
    return true;
}

int K_SafeTensors::OpenDir(const std::string& dir) {
    const std::string single = dir + "/model.safetensors", idx = dir + "/model.safetensors.index.json";
    if (file_exists(single)) return OpenFile(single);
    std::string text;
    if (!read_text(idx, text)) {
        err = "neither model.safetensors nor model.safetensors.index.json under " + dir;
        return KF_INVALID_ARGS;
    }
    JSON j;
    std::string jerr;
    if (!JSON::Parse(text.data(), text.size(), j, jerr)) {
        err = idx + ": " + jerr;
        return KF_INVALID_ARGS;
    }
    const JSON* wm = j.get("weight_map");
    if (!wm || wm->kind != JSON::OBJ) {
        err = idx + ": no weight_map";
        return KF_INVALID_ARGS;
    }
    std::vector<std::string> shards;
    for (const auto& kv : wm->obj) {
        bool seen = false;
        for (const auto& s : shards) seen = seen || s == kv.second.str;
        if (!seen) shards.push_back(kv.second.str);
    }
    for (const auto& s : shards) {
        int rc = OpenFile(dir + "/" + s);
        if (rc != KF_OK) return rc;
    }
    return KF_OK;
}

const ST_Tensor* K_SafeTensors::Find(const std::string& name) const {
    auto it = index.find(name);
    return it == index.end() ? nullptr : &tensors[it->second];
}
const void* K_SafeTensors::Data(const ST_Tensor& t) const {
    const File& f = files[t.file];
    return reinterpret_cast<const unsigned char*>(f.map) + f.data_base + t.begin;
}

// ------------------------------------------------------------------------------------------------ HF checkpoint -> Fish
namespace {

inline uint16_t f32_to_bf16(float f) {  // round to nearest even, NaN kept
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline float half_to_f32(uint16_t h) {
    const uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
    uint32_t u;
    if (e == 0) {
        if (m == 0) {
            u = s;
        } else {
            int sh = 0;
            uint32_t mm = m;
            while (!(mm & 1024u)) mm <<= 1, sh++;
            u = s | ((uint32_t)(113 - sh) << 23) | ((mm & 1023u) << 13);
        }
    } else if (e == 31) {
        u = s | 0x7f800000u | (m << 13);
    } else {
        u = s | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}

void quant_range(typNUMBER t, bool symmetric, QuantCard& q) {  // GeQuant ctor (GeQuant.cpp:107-124)
    q.qMin = q.qMax = q.qBias = 0, q.bits = 16;
    switch (t) {
        case typNUMBER::Q4: q.bits = 4; if (symmetric) q.qMin = -8, q.qMax = 7, q.qBias = 8; else q.qMin = 0, q.qMax = 15; break;
        case typNUMBER::T_SIGN: q.bits = 2, q.qMin = -1, q.qMax = 1, q.qBias = 1; break;
        case typNUMBER::BOOL1: case typNUMBER::T_BINARY: q.bits = 1, q.qMin = 0, q.qMax = 1; break;
        case typNUMBER::F8E5M2: q.bits = 8; break;
        default: break;
    }
}

struct Loader {
    Fish* f;
    K_SafeTensors& st;
    std::string& err;
    void* d_stage = nullptr;  // bf16 staging on the device, grown on demand
    size_t stage_bytes = 0;
    std::vector<uint16_t> h_conv;

    int fail(int code, const std::string& what) {
        err = what;
        return code;
    }
    ~Loader() {
        if (d_stage) kf_free(f->ctx, d_stage);
    }
    // bf16 [ne0, ne1] of tensor `name` on the device (staging buffer); F16 / F32 sources are converted on the host
    int stage(const std::string& name, int ne0, int ne1, const uint16_t** d_out) {
        const ST_Tensor* t = st.Find(name);
        if (!t) return fail(KF_INVALID_ARGS, "tensor '" + name + "' not in the checkpoint");
        const bool shape_ok = (t->shape.size() == 2 && t->shape[0] == ne0 && t->shape[1] == ne1) || (t->shape.size() == 1 && ne1 == 1 && t->shape[0] == ne0);
        if (!shape_ok) return fail(KF_INVALID_ARGS, "tensor '" + name + "' has an unexpected shape");
        const size_t n = (size_t)ne0 * ne1;
        const void* src = st.Data(*t);
        if (t->dtype == "F32") {
            h_conv.resize(n);
            const float* p = reinterpret_cast<const float*>(src);
            for (size_t i = 0; i < n; i++) h_conv[i] = f32_to_bf16(p[i]);
            src = h_conv.data();
        } else if (t->dtype == "F16") {
            h_conv.resize(n);
            const uint16_t* p = reinterpret_cast<const uint16_t*>(src);
            for (size_t i = 0; i < n; i++) h_conv[i] = f32_to_bf16(half_to_f32(p[i]));
            src = h_conv.data();
        } else if (t->dtype != "BF16") {
            return fail(KF_UNSUPPORTED_DATATYPE, "tensor '" + name + "': dtype " + t->dtype + " (BF16, F16 or F32 expected)");
        }
        if (n * 2 > stage_bytes) {
            if (d_stage) kf_free(f->ctx, d_stage);
            d_stage = nullptr;
            KF_TRY(kf_malloc(f->ctx, n * 2, &d_stage));
            stage_bytes = n * 2;
        }
        KF_TRY(kf_h2d(f->ctx, d_stage, src, n * 2));
        *d_out = reinterpret_cast<const uint16_t*>(d_stage);
        return KF_OK;
    }
    // dense HF weight -> GTensor of type `tp` (quantise-on-load, GeQuant.cpp:144-200 -> RTN_x / YinYang on the device)
    int dense(const std::string& name, int ne0, int ne1, typNUMBER tp, int lGroup, hGTensor* out) {
        const uint16_t* d_src = nullptr;
        KF_TRY(stage(name, ne0, ne1, &d_src));
        auto t = std::make_shared<GTensor>();
        t->name = name, t->type = tp, t->ne[0] = ne0, t->ne[1] = ne1;
        quant_range(tp, false, t->quant);
        t->quant.T_group = lGroup;
        const size_t n = (size_t)ne0 * ne1;
        if (tp == typNUMBER::BF16) {
            t->szData = n * 2;
            KF_TRY(t->Alloc(f->ctx, t->szData));
            KF_TRY(kf_d2d(f->ctx, t->data, d_src, t->szData));
        } else if (tp == typNUMBER::F8E5M2) {
            t->szData = n;
            KF_TRY(t->Alloc(f->ctx, t->szData));
            kf_weight d = t->desc();
            KF_TRY(kf_quantize(f->ctx, &d, d_src, 0));
        } else if (tp == typNUMBER::Q4 && lGroup == 0) {
            // quant card with isNormalFloat (QUANT_MODE::RTNf): GeQuant::RT_NormalF on the device -> nibble stream || [R][C][LUT ne0 x 16]
            if ((ne1 % 32) || (ne0 % 8)) return fail(KF_QUANT_ERR, "tensor '" + name + "': the normal-float row form needs in % 32 == 0 and out % 8 == 0");
            t->quant.isNormalFloat = true;
            t->szData = n / 2;
            t->szGama = ((size_t)ne0 + ne1 + 16 * (size_t)ne0) * 2;
            KF_TRY(t->Alloc(f->ctx, t->szData + t->szGama));
            std::vector<uint16_t> ones((size_t)ne0 + ne1, 0x3F80);
            KF_TRY(kf_h2d(f->ctx, t->gama_T(), ones.data(), ones.size() * 2));
            kf_weight d = t->desc();
            KF_TRY(kf_quantize(f->ctx, &d, d_src, 0));
        } else {
            if (lGroup <= 0) return fail(KF_QUANT_ERR, "tensor '" + name + "': group size " + std::to_string(lGroup));
            if (n % (size_t)lGroup) return fail(KF_QUANT_ERR, "tensor '" + name + "': size is not a multiple of the group " + std::to_string(lGroup));
            t->szData = n * t->quant.bits / 8;
            const size_t nGroup = n / lGroup;
            t->szGama = ((size_t)ne0 + ne1 + 2 * nGroup) * 2;
            KF_TRY(t->Alloc(f->ctx, t->szData + t->szGama));
            std::vector<uint16_t> ones((size_t)ne0 + ne1, 0x3F80);  // R_SCALE / C_SCALE = 1 (rc_normal = 0)
            KF_TRY(kf_h2d(f->ctx, t->gama_T(), ones.data(), ones.size() * 2));
            kf_weight d = t->desc();
            KF_TRY(kf_quantize(f->ctx, &d, d_src, 0));
        }
        *out = t;
        return KF_OK;
    }
    int raw_copy(const ST_Tensor& s, hGTensor* out, typNUMBER tp) {
        auto t = std::make_shared<GTensor>();
        t->name = s.name, t->type = tp, t->szData = s.end - s.begin;
        KF_TRY(t->Alloc(f->ctx, t->szData));
        KF_TRY(kf_h2d(f->ctx, t->data, st.Data(s), t->szData));
        *out = t;
        return KF_OK;
    }
    // vendor AutoAWQ linear (GeQuant.cpp:410, CU_Q42X_awq): <prefix>.qweight I32 [in, out/8], .qzeros I32 [in/128, out/8], .scales F16 [in/128, out]
    int awq(const std::string& prefix, int n_out, int n_in, hGTensor* out) {
        const ST_Tensor *qw = st.Find(prefix + ".qweight"), *qz = st.Find(prefix + ".qzeros"), *sc = st.Find(prefix + ".scales");
        if (!qw || !qz || !sc) return fail(KF_INVALID_ARGS, prefix + ": qweight/qzeros/scales incomplete");
        const bool ok = qw->dtype == "I32" && qz->dtype == "I32" && sc->dtype == "F16" && qw->shape.size() == 2 && qw->shape[0] == n_in && qw->shape[1] == n_out / 8 &&
                        qz->shape.size() == 2 && qz->shape[0] == n_in / 128 && qz->shape[1] == n_out / 8 && sc->shape.size() == 2 && sc->shape[0] == n_in / 128 &&
                        sc->shape[1] == n_out;
        if (!ok) return fail(KF_QUANT_ERR, prefix + ": not the AutoAWQ GEMM layout with group 128");
        hGTensor t, z, s;
        KF_TRY(raw_copy(*qw, &t, typNUMBER::Q4));
        KF_TRY(raw_copy(*qz, &z, typNUMBER::I32));
        KF_TRY(raw_copy(*sc, &s, typNUMBER::F16));
        t->ne[0] = n_out, t->ne[1] = n_in, t->quant.bits = 4, t->quant.T_group = 128, t->quant.qMin = 0, t->quant.qMax = 15, t->quant.qBias = 0;
        t->qZero = z, t->qScale = s;
        *out = t;
        return KF_OK;
    }
    int linear(const std::string& prefix, int n_out, int n_in, typNUMBER tp, int lGroup, SLP* slot, bool* any_awq) {
        hGTensor t;
        if (st.Find(prefix + ".qweight")) {
            KF_TRY(awq(prefix, n_out, n_in, &t));
            *any_awq = true;
        } else {
            KF_TRY(dense(prefix + ".weight", n_out, n_in, tp, lGroup, &t));
        }
        slot->w = t, slot->nOut = n_out, slot->nIn = n_in;
        return slot->hFish ? slot->hFish->EnsureLinearScratch(t->desc(), 1) : KF_OK; /* AutoAWQ tensors: the mat-vec's slice partials live in caller-owned scratch */
    }
    int norm(const std::string& name, int n, LayerNormal* ln, bool required) {
        if (!st.Find(name)) return required ? fail(KF_INVALID_ARGS, "tensor '" + name + "' not in the checkpoint") : KF_OK;
        hGTensor t;
        KF_TRY(dense(name, n, 1, typNUMBER::BF16, 128, &t));
        ln->w = t;
        return KF_OK;
    }
};

}  // namespace

// HF directory -> Fish.  layer_type / head_type: what the dense matrices are quantised to on load (BF16 keeps them).
Fish* LoadHF(const std::string& dir, int device, void* stream, typNUMBER layer_type, typNUMBER head_type, int lGroup, int max_seq, int* rc_out, std::string& err,
             bool layer_nf, bool head_nf) {
    const int lGroupL = layer_nf ? 0 : lGroup, lGroupH = head_nf ? 0 : lGroup; /* 0: normal-float row codebooks (Loader::dense) */
    auto bail = [&](int rc, const std::string& what) -> Fish* {
        if (rc_out) *rc_out = rc;
        err = what;
        return nullptr;
    };
    std::string text;
    if (!read_text(dir + "/config.json", text)) return bail(KF_INVALID_ARGS, "cannot read " + dir + "/config.json");
    JSON cfg;
    std::string jerr;
    if (!JSON::Parse(text.data(), text.size(), cfg, jerr) || cfg.kind != JSON::OBJ) return bail(KF_INVALID_ARGS, "config.json: " + jerr);
    MODEL_CARD card;  // CLI_params.cpp:2177-2300
    card.nEmbed = (int)cfg.number_or("hidden_size", 0);
    card.nLayer = (int)cfg.number_or("num_hidden_layers", 0);
    card.n_head = (int)cfg.number_or("num_attention_heads", 0);
    card.n_head_kv = (int)cfg.number_or("num_key_value_heads", card.n_head);
    card.head_dim = (int)cfg.number_or("head_dim", card.n_head ? card.nEmbed / card.n_head : 0);
    card.n_ff = (int)cfg.number_or("intermediate_size", 0);
    card.vocab = (int)cfg.number_or("vocab_size", 0);
    card.rms_eps = card.qk_eps = (float)cfg.number_or("rms_norm_eps", 1e-6);
    card.rope_theta = (float)cfg.number_or("rope_theta", 10000.0);  // Neuron.cpp:612-624 falls back to 10000 when the card has none
    card.tie_word_embeddings = cfg.bool_or("tie_word_embeddings", false);
    const int max_pos = (int)cfg.number_or("max_position_embeddings", 2048);
    card.n_ctx = max_seq > 0 ? max_seq : (max_pos < 4096 ? max_pos : 4096);
    if (card.nEmbed <= 0 || card.nLayer <= 0 || card.n_head <= 0 || card.n_head_kv <= 0 || card.head_dim <= 0 || card.n_ff <= 0 || card.vocab <= 0)
        return bail(KF_INVALID_ARGS, "config.json lacks hidden_size / num_hidden_layers / num_attention_heads / intermediate_size / vocab_size");

    K_SafeTensors st;
    int rc = st.OpenDir(dir);
    if (rc != KF_OK) return bail(rc, st.err);

    std::unique_ptr<Fish> f(new Fish());
    rc = f->Build(card, device, stream);
    if (rc != KF_OK) return bail(rc, "Fish::Build failed");
    Loader L{f.get(), st, err};
    const int C = card.nEmbed, qd = card.n_head * card.head_dim, kvd = card.n_head_kv * card.head_dim;
    bool any_awq = false;
    auto check = [&](int r) { return r == KF_OK; };
    hGTensor emb;
    if (!check(rc = L.dense("model.embed_tokens.weight", card.vocab, C, head_type, lGroupH, &emb))) return bail(rc, err);
    f->embed.w = emb;
    if (card.tie_word_embeddings || !st.Find("lm_head.weight")) {
        f->head.proj.w = emb, f->head.proj.nOut = card.vocab, f->head.proj.nIn = C;  // Neuron.cpp:349-356
    } else {
        hGTensor hw;
        if (!check(rc = L.dense("lm_head.weight", card.vocab, C, head_type, lGroupH, &hw))) return bail(rc, err);
        f->head.proj.w = hw, f->head.proj.nOut = card.vocab, f->head.proj.nIn = C;
    }
    if (!check(rc = L.norm("model.norm.weight", C, &f->final_norm, true))) return bail(rc, err);
    for (int l = 0; l < card.nLayer; l++) {
        const std::string p = "model.layers." + std::to_string(l) + ".";
        SelfAttention* a = f->attn[l].get();
        FFN* m = f->ffn[l].get();
        if (!check(rc = L.norm(p + "input_layernorm.weight", C, &a->norm, true)) || !check(rc = L.norm(p + "post_attention_layernorm.weight", C, &m->norm, true)) ||
            !check(rc = L.norm(p + "self_attn.q_norm.weight", card.head_dim, &a->normQ, false)) ||
            !check(rc = L.norm(p + "self_attn.k_norm.weight", card.head_dim, &a->normK, false)) ||
            !check(rc = L.linear(p + "self_attn.q_proj", qd, C, layer_type, lGroupL, &a->Q, &any_awq)) ||
            !check(rc = L.linear(p + "self_attn.k_proj", kvd, C, layer_type, lGroupL, &a->K, &any_awq)) ||
            !check(rc = L.linear(p + "self_attn.v_proj", kvd, C, layer_type, lGroupL, &a->V, &any_awq)) ||
            !check(rc = L.linear(p + "self_attn.o_proj", C, qd, layer_type, lGroupL, &a->proj_cat, &any_awq)) ||
            !check(rc = L.linear(p + "mlp.gate_proj", card.n_ff, C, layer_type, lGroupL, &m->gate, &any_awq)) ||
            !check(rc = L.linear(p + "mlp.up_proj", card.n_ff, C, layer_type, lGroupL, &m->up, &any_awq)) ||
            !check(rc = L.linear(p + "mlp.down_proj", C, card.n_ff, layer_type, lGroupL, &m->down, &any_awq)))
            return bail(rc, err);
    }
    if (any_awq) f->fuse_level = 0;  // the AutoAWQ layout has its own mat-vec (kf_linear only): one launch per reference kernel
    if (rc_out) *rc_out = KF_OK;
    return f.release();
}

}  // namespace koifish

// ================================================================================================ C entry points (ctypes)
using namespace koifish;
static thread_local std::string g_st_err;
extern "C" {

const char* kfh_last_error(void) { return g_st_err.c_str(); }

// checkpoint inspection without a GPU (tests): open, count, describe, close
void* kfh_st_open(const char* path_or_dir, int is_dir) {
    auto* st = new K_SafeTensors();
    const int rc = is_dir ? st->OpenDir(path_or_dir) : st->OpenFile(path_or_dir);
    if (rc != KF_OK) {
        g_st_err = st->err;
        delete st;
        return nullptr;
    }
    return st;
}
void kfh_st_close(void* h) { delete reinterpret_cast<K_SafeTensors*>(h); }
int kfh_st_count(void* h) { return (int)reinterpret_cast<K_SafeTensors*>(h)->tensors.size(); }
// name/dtype copied into caller buffers (64 / 16 bytes suffice for HF names... name_cap given), shape up to 4 dims, offsets
int kfh_st_info(void* h, int i, char* name, int name_cap, char* dtype, int dtype_cap, int64_t* shape4, int* ndim, uint64_t* begin, uint64_t* end) {
    auto* st = reinterpret_cast<K_SafeTensors*>(h);
    if (i < 0 || i >= (int)st->tensors.size()) return KF_INVALID_ARGS;
    const ST_Tensor& t = st->tensors[i];
    snprintf(name, name_cap, "%s", t.name.c_str());
    snprintf(dtype, dtype_cap, "%s", t.dtype.c_str());
    *ndim = (int)t.shape.size();
    for (int d = 0; d < 4; d++) shape4[d] = d < (int)t.shape.size() ? t.shape[d] : 0;
    *begin = t.begin, *end = t.end;
    return KF_OK;
}
// first bytes of a tensor's data (tests compare them with the writer's)
int kfh_st_read(void* h, const char* name, void* out, uint64_t nbytes) {
    auto* st = reinterpret_cast<K_SafeTensors*>(h);
    const ST_Tensor* t = st->Find(name);
    if (!t || nbytes > t->end - t->begin) return KF_INVALID_ARGS;
    memcpy(out, st->Data(*t), nbytes);
    return KF_OK;
}

// HF directory (config.json + model.safetensors[.index.json]) -> Fish handle usable with every kfh_* entry; NULL + *rc on failure
void* kfh_load_hf(const char* dir, int device, void* stream, int layer_type, int head_type, int lGroup, int max_seq, int* rc) {
    std::string err;
    // 1000 (koifish_amd.lib.NF4) is not a typNUMBER: "Q4 with the normal-float quant card" (QUANT_MODE::RTNf)
    const bool lnf = layer_type == 1000, hnf = head_type == 1000;
    Fish* f = LoadHF(dir, device, stream, lnf ? typNUMBER::Q4 : (typNUMBER)layer_type, hnf ? typNUMBER::Q4 : (typNUMBER)head_type, lGroup > 0 ? lGroup : 128, max_seq, rc, err,
                     lnf, hnf);
    if (!f) g_st_err = err;
    return f;
}
// {dim, n_layer, n_head, n_kv, head_dim, ffn, vocab, n_ctx, tied, fuse_level} and {rms_eps, rope_theta}
int kfh_get_config(void* h, int* out10, float* out2) {
    Fish* f = reinterpret_cast<Fish*>(h);
    const MODEL_CARD& c = f->config;
    const int v[10] = {c.nEmbed, c.nLayer, c.n_head, c.n_head_kv, c.head_dim, c.n_ff, c.vocab, c.n_ctx, c.tie_word_embeddings ? 1 : 0, f->fuse_level};
    memcpy(out10, v, sizeof(v));
    out2[0] = c.rms_eps, out2[1] = c.rope_theta;
    return KF_OK;
}
}
