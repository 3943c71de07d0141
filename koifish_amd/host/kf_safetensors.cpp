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
            v.is_int = !strpbrk(buf, ".eE");
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
JSON JSON::Str(const std::string& s) {
    JSON v;
    v.kind = STR, v.str = s;
    return v;
}
JSON JSON::Int(int64_t x) {
    JSON v;
    v.kind = NUM, v.num = (double)x, v.is_int = true;
    return v;
}
JSON JSON::Real(double x) {
    JSON v;
    v.kind = NUM, v.num = x;
    return v;
}
JSON JSON::Bool(bool x) {
    JSON v;
    v.kind = BOOL, v.b = x;
    return v;
}
JSON JSON::Object() {
    JSON v;
    v.kind = OBJ;
    return v;
}
JSON JSON::Array() {
    JSON v;
    v.kind = ARR;
    return v;
}
JSON& JSON::operator[](const std::string& key) {
    if (kind == NUL) kind = OBJ;
    for (auto& kv : obj)
        if (kv.first == key) return kv.second;
    obj.emplace_back(key, JSON());
    return obj.back().second;
}
const JSON* JSON::path(std::initializer_list<const char*> keys) const {
    const JSON* v = this;
    for (const char* k : keys) {
        v = v->get(k);
        if (!v) return nullptr;
    }
    return v;
}

namespace {
void dump_string(const std::string& s, std::string& out) {
    out += '"';
    for (unsigned char c : s) {
        switch (c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\n': out += "\\n"; break;
            case '\t': out += "\\t"; break;
            case '\r': out += "\\r"; break;
            case '\b': out += "\\b"; break;
            case '\f': out += "\\f"; break;
            default:
                if (c < 0x20) {
                    char buf[8];
                    snprintf(buf, sizeof(buf), "\\u%04x", c);
                    out += buf;
                } else {
                    out += (char)c;
                }
        }
    }
    out += '"';
}
void dump_value(const JSON& v, std::string& out) {
    char buf[40];
    switch (v.kind) {
        case JSON::NUL: out += "null"; break;
        case JSON::BOOL: out += v.b ? "true" : "false"; break;
        case JSON::NUM:
            if (v.is_int) {
                snprintf(buf, sizeof(buf), "%lld", (long long)v.num);
            } else { /* shortest text that reads back to the same double */
                for (int prec = 1; prec <= 17; prec++) {
                    snprintf(buf, sizeof(buf), "%.*g", prec, v.num);
                    if (strtod(buf, nullptr) == v.num) break;
                }
                if (!strpbrk(buf, ".eEn")) strcat(buf, ".0"); /* 2.0 stays a real */
            }
            out += buf;
            break;
        case JSON::STR: dump_string(v.str, out); break;
        case JSON::ARR:
            out += '[';
            for (size_t i = 0; i < v.arr.size(); i++) {
                if (i) out += ',';
                dump_value(v.arr[i], out);
            }
            out += ']';
            break;
        case JSON::OBJ:
            out += '{';
            for (size_t i = 0; i < v.obj.size(); i++) {
                if (i) out += ',';
                dump_string(v.obj[i].first, out);
                out += ':';
                dump_value(v.obj[i].second, out);
            }
            out += '}';
            break;
    }
}
void be(std::vector<uint8_t>& o, uint64_t v, int nbytes) {
    for (int i = nbytes - 1; i >= 0; i--) o.push_back((uint8_t)(v >> (8 * i)));
}
void mp_str(const std::string& s, std::vector<uint8_t>& o) {
    const size_t n = s.size();
    if (n <= 31) o.push_back((uint8_t)(0xa0 | n));
    else if (n <= 0xff) o.push_back(0xd9), be(o, n, 1);
    else if (n <= 0xffff) o.push_back(0xda), be(o, n, 2);
    else o.push_back(0xdb), be(o, n, 4);
    o.insert(o.end(), s.begin(), s.end());
}
void mp_value(const JSON& v, std::vector<uint8_t>& o) {
    switch (v.kind) {
        case JSON::NUL: o.push_back(0xc0); break;
        case JSON::BOOL: o.push_back(v.b ? 0xc3 : 0xc2); break;
        case JSON::NUM:
            if (v.is_int) {
                const int64_t i = (int64_t)v.num;
                if (i >= 0) {
                    const uint64_t u = (uint64_t)i;
                    if (u <= 0x7f) o.push_back((uint8_t)u);
                    else if (u <= 0xff) o.push_back(0xcc), be(o, u, 1);
                    else if (u <= 0xffff) o.push_back(0xcd), be(o, u, 2);
                    else if (u <= 0xffffffffull) o.push_back(0xce), be(o, u, 4);
                    else o.push_back(0xcf), be(o, u, 8);
                } else if (i >= -32) {
                    o.push_back((uint8_t)(int8_t)i);
                } else if (i >= -128) {
                    o.push_back(0xd0), be(o, (uint64_t)i, 1);
                } else if (i >= -32768) {
                    o.push_back(0xd1), be(o, (uint64_t)i, 2);
                } else if (i >= -2147483648ll) {
                    o.push_back(0xd2), be(o, (uint64_t)i, 4);
                } else {
                    o.push_back(0xd3), be(o, (uint64_t)i, 8);
                }
            } else {
                const float f = (float)v.num;
                if (std::fabs(v.num) <= 3.4028234663852886e38 && (double)f == v.num) {
                    uint32_t u;
                    memcpy(&u, &f, 4);
                    o.push_back(0xca), be(o, u, 4);
                } else {
                    uint64_t u;
                    memcpy(&u, &v.num, 8);
                    o.push_back(0xcb), be(o, u, 8);
                }
            }
            break;
        case JSON::STR: mp_str(v.str, o); break;
        case JSON::ARR: {
            const size_t n = v.arr.size();
            if (n <= 15) o.push_back((uint8_t)(0x90 | n));
            else if (n <= 0xffff) o.push_back(0xdc), be(o, n, 2);
            else o.push_back(0xdd), be(o, n, 4);
            for (const auto& c : v.arr) mp_value(c, o);
            break;
        }
        case JSON::OBJ: {
            const size_t n = v.obj.size();
            if (n <= 15) o.push_back((uint8_t)(0x80 | n));
            else if (n <= 0xffff) o.push_back(0xde), be(o, n, 2);
            else o.push_back(0xdf), be(o, n, 4);
            for (const auto& kv : v.obj) mp_str(kv.first, o), mp_value(kv.second, o);
            break;
        }
    }
}
struct MpReader {
    const uint8_t *p, *end, *base;
    std::string err;
    bool fail(const char* what) {
        char buf[96];
        snprintf(buf, sizeof(buf), "msgpack: %s at byte %zu", what, (size_t)(p - base));
        err = buf;
        return false;
    }
    bool need(size_t n) { return (size_t)(end - p) >= n ? true : fail("truncated"); }
    uint64_t rd(int nbytes) {
        uint64_t v = 0;
        for (int i = 0; i < nbytes; i++) v = (v << 8) | *p++;
        return v;
    }
    bool str(size_t n, std::string& out) {
        if (!need(n)) return false;
        out.assign(reinterpret_cast<const char*>(p), n);
        p += n;
        return true;
    }
    bool arr(size_t n, JSON& v, int depth) {
        v.kind = JSON::ARR;
        if (n > (size_t)(end - p)) return fail("array longer than the buffer");
        v.arr.resize(n);
        for (size_t i = 0; i < n; i++)
            if (!value(v.arr[i], depth + 1)) return false;
        return true;
    }
    bool map(size_t n, JSON& v, int depth) {
        v.kind = JSON::OBJ;
        if (n > (size_t)(end - p)) return fail("map longer than the buffer");
        v.obj.resize(n);
        for (size_t i = 0; i < n; i++) {
            JSON k;
            if (!value(k, depth + 1)) return false;
            if (k.kind != JSON::STR) return fail("map key is not a string");
            v.obj[i].first = std::move(k.str);
            if (!value(v.obj[i].second, depth + 1)) return false;
        }
        return true;
    }
    bool value(JSON& v, int depth) {
        if (depth > 64) return fail("nesting too deep");
        if (!need(1)) return false;
        const uint8_t t = *p++;
        auto integer = [&](int64_t x) {
            v.kind = JSON::NUM, v.is_int = true, v.num = (double)x;
            return true;
        };
        if (t <= 0x7f) return integer(t);
        if (t >= 0xe0) return integer((int8_t)t);
        if ((t & 0xf0) == 0x80) return map(t & 15, v, depth);
        if ((t & 0xf0) == 0x90) return arr(t & 15, v, depth);
        if ((t & 0xe0) == 0xa0) {
            v.kind = JSON::STR;
            return str(t & 31, v.str);
        }
        switch (t) {
            case 0xc0: v.kind = JSON::NUL; return true;
            case 0xc2: v.kind = JSON::BOOL, v.b = false; return true;
            case 0xc3: v.kind = JSON::BOOL, v.b = true; return true;
            case 0xc4: case 0xc5: case 0xc6: case 0xd9: case 0xda: case 0xdb: { /* bin 8/16/32 (kept as a string) and str 8/16/32 */
                const int lb = (t == 0xc4 || t == 0xd9) ? 1 : (t == 0xc5 || t == 0xda) ? 2 : 4;
                if (!need(lb)) return false;
                v.kind = JSON::STR;
                return str((size_t)rd(lb), v.str);
            }
            case 0xca: {
                if (!need(4)) return false;
                const uint32_t u = (uint32_t)rd(4);
                float f;
                memcpy(&f, &u, 4);
                v.kind = JSON::NUM, v.num = f;
                return true;
            }
            case 0xcb: {
                if (!need(8)) return false;
                const uint64_t u = rd(8);
                memcpy(&v.num, &u, 8);
                v.kind = JSON::NUM;
                return true;
            }
            case 0xcc: return need(1) && integer((int64_t)rd(1));
            case 0xcd: return need(2) && integer((int64_t)rd(2));
            case 0xce: return need(4) && integer((int64_t)rd(4));
            case 0xcf: return need(8) && integer((int64_t)rd(8));
            case 0xd0: return need(1) && integer((int8_t)rd(1));
            case 0xd1: return need(2) && integer((int16_t)rd(2));
            case 0xd2: return need(4) && integer((int32_t)rd(4));
            case 0xd3: return need(8) && integer((int64_t)rd(8));
            case 0xdc: return need(2) && arr((size_t)rd(2), v, depth);
            case 0xdd: return need(4) && arr((size_t)rd(4), v, depth);
            case 0xde: return need(2) && map((size_t)rd(2), v, depth);
            case 0xdf: return need(4) && map((size_t)rd(4), v, depth);
        }
        return fail("unsupported type byte");
    }
};
}  // namespace

std::string JSON::Dump() const {
    std::string out;
    dump_value(*this, out);
    return out;
}
void JSON::ToMsgpack(std::vector<uint8_t>& out) const { mp_value(*this, out); }
bool JSON::FromMsgpack(const uint8_t* p, size_t n, JSON& out, std::string& err) {
    MpReader r{p, p + n, p, {}};
    out = JSON();
    if (!r.value(out, 0)) {
        err = r.err;
        return false;
    }
    if (r.p != r.end) {
        r.fail("trailing bytes");
        err = r.err;
        return false;
    }
    return true;
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
        bool dims_ok = off->arr[0].kind == JSON::NUM && off->arr[1].kind == JSON::NUM && off->arr[0].num >= 0 && off->arr[1].num >= 0 && off->arr[0].num < 9.0e15 &&
                       off->arr[1].num < 9.0e15;
        for (const auto& d : sh->arr) { /* untrusted input: every dimension a non-negative integer, the element count without overflow */
            if (d.kind != JSON::NUM || d.num < 0 || d.num > 1.0e15 || d.num != std::floor(d.num)) {
                dims_ok = false;
                break;
            }
            const size_t v = (size_t)d.num;
            if (v != 0 && count > ((size_t)1 << 60) / v) dims_ok = false;
            t.shape.push_back((int64_t)v);
            count *= v;
        }
        if (!dims_ok) {
            munmap(f.map, f.size), close(f.fd);
            err = path + ": tensor entry '" + kv.first + "' has a malformed shape or data_offsets";
            return KF_INVALID_ARGS;
        }
        t.begin = (size_t)off->arr[0].num, t.end = (size_t)off->arr[1].num;
        t.szData = (size_t)kv.second.number_or("szData", 0), t.szGama = (size_t)kv.second.number_or("szGama", 0);
        const bool kun = t.szData + t.szGama > 0; /* GTensor::jDesc (Serialize.cpp:92-99): the entry spans data||gama, whatever dtype says */
        const size_t eb = kun ? 0 : dtype_bytes(t.dtype);
        if (t.begin > t.end || t.end > avail || (eb && t.end - t.begin != count * eb) || (kun && t.end - t.begin != t.szData + t.szGama)) {
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

bool K_SafeTensors::Config(JSON& out, std::string& e) const {
    const ST_Tensor* t = Find(config_key());
    if (!t) {
        e = "no " + std::string(config_key()) + " tensor";
        return false;
    }
    return JSON::FromMsgpack(reinterpret_cast<const uint8_t*>(Data(*t)), t->end - t->begin, out, e);
}

// ------------------------------------------------------------------------------------------------ .kun writer
// typNUMBER <-> K_FLOATS name (src/g_float.hpp:127-151); tpNumOf upper-cases and also accepts the aliases (GST_float.cpp:22-43)
static const struct {
    typNUMBER t;
    const char *name, *alias;
} kFloats[] = {{typNUMBER::F32, "FLOAT", "F32"},   {typNUMBER::F16, "F16(E5)", "F16"}, {typNUMBER::U8, "U8", nullptr},       {typNUMBER::I8, "I8", nullptr},
               {typNUMBER::U16, "U16", nullptr},   {typNUMBER::I16, "I16", nullptr},   {typNUMBER::U32, "U32", nullptr},     {typNUMBER::I32, "I32", nullptr},
               {typNUMBER::U64, "U64", nullptr},   {typNUMBER::I64, "I64", nullptr},   {typNUMBER::F64, "F64", nullptr},     {typNUMBER::BF16, "BF16(E8)", "BF16"},
               {typNUMBER::F8E5M2, "F8E5M2", nullptr}, {typNUMBER::F8E4M3, "F8E4M3", nullptr}, {typNUMBER::Q4, "Q<4>", nullptr}, {typNUMBER::Q3, "Q<3>", nullptr},
               {typNUMBER::Q2, "Q<2>", nullptr},   {typNUMBER::T_SIGN, "TERNARY", nullptr}, {typNUMBER::BOOL1, "BOOL<1>", nullptr}, {typNUMBER::T_BINARY, "BINARY", nullptr}};
const char* K_FLOATS_name(int typ) {
    for (const auto& k : kFloats)
        if ((int)k.t == typ) return k.name;
    return nullptr;
}
int K_FLOATS_type(const std::string& name) {
    std::string u = name;
    for (auto& c : u) c = (char)toupper((unsigned char)c);
    for (const auto& k : kFloats)
        if (u == k.name || (k.alias && u == k.alias)) return (int)k.t;
    return -1;
}

size_t KunWriter::Register(const std::string& name, const std::string& dtype, const std::vector<int64_t>& shape, size_t szData, size_t szGama) {
    Entry e;
    e.name = name, e.dtype = dtype, e.shape = shape, e.szData = szData, e.szGama = szGama, e.begin = offset;
    offset += szData + szGama;  // nByte_CKP (Serialize.cpp:226-250): nByte() + szGama for a checkpoint that is not a training state
    entries.push_back(std::move(e));
    return offset;
}

static JSON kun_desc(const std::string& dtype, const std::vector<int64_t>& shape, size_t b, size_t e, size_t szGama, size_t szData) {  // GTensor::jDesc
    JSON d = JSON::Object();
    d["dtype"] = JSON::Str(dtype);
    JSON sh = JSON::Array();
    for (int64_t x : shape) sh.arr.push_back(JSON::Int(x));
    d["shape"] = sh;
    JSON off = JSON::Array();
    off.arr.push_back(JSON::Int((int64_t)b)), off.arr.push_back(JSON::Int((int64_t)e));
    d["data_offsets"] = off;
    d["loAB"] = JSON::Int(0);
    d["szGama"] = JSON::Int((int64_t)szGama);
    d["szData"] = JSON::Int((int64_t)szData);
    return d;
}

static bool write_all(int fd, const void* p, size_t n) {
    const char* c = reinterpret_cast<const char*>(p);
    while (n) {
        const ssize_t w = write(fd, c, n);
        if (w <= 0) return false;
        c += w, n -= (size_t)w;
    }
    return true;
}

int KunWriter::Save(const std::string& path, JSON jsConfig, const std::function<int(size_t, void*, size_t)>& fetch, std::string& err) {
    JSON& jt = jsConfig["tensors"];
    if (jt.kind == JSON::NUL) jt = JSON::Object();
    for (const auto& e : entries) jt[e.name] = JSON::Int((int64_t)e.begin);
    std::vector<uint8_t> pack;
    jsConfig.ToMsgpack(pack);  // insertJS: the config rides as the last tensor, dtype U8, shape [bytes]
    JSON hdr = JSON::Object();
    JSON meta = JSON::Object();
    meta["format"] = JSON::Str("pt"), meta["writer"] = JSON::Str("koifish");  // K_SafeTensors::UpdateMetaData (Serialize.cpp:842-847)
    hdr["__metadata__"] = meta;
    for (const auto& e : entries) hdr[e.name] = kun_desc(e.dtype, e.shape, e.begin, e.begin + e.szData + e.szGama, e.szGama, e.szData);
    hdr[K_SafeTensors::config_key()] = kun_desc("U8", {(int64_t)pack.size()}, offset, offset + pack.size(), 0, 0);
    const std::string text = hdr.Dump();
    const std::string tmp = path + ".tmp";
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd < 0) {
        err = "cannot create " + tmp;
        return KF_INVALID_ARGS;
    }
    auto bail = [&](int rc, const std::string& what) {
        close(fd), unlink(tmp.c_str());
        err = what;
        return rc;
    };
    const uint64_t hlen = text.size();  // little-endian u64 (this host is), no padding (_to_ofs: `if (0)` around the 8-byte pad)
    if (!write_all(fd, &hlen, 8) || !write_all(fd, text.data(), text.size())) return bail(KF_INTERNAL_ERR, "write failed: " + tmp);
    std::vector<uint8_t> buf;
    for (size_t i = 0; i < entries.size(); i++) {
        const size_t n = entries[i].szData + entries[i].szGama;
        buf.resize(n);
        const int rc = fetch(i, buf.data(), n);
        if (rc != KF_OK) return bail(rc, "reading tensor '" + entries[i].name + "' failed");
        if (!write_all(fd, buf.data(), n)) return bail(KF_INTERNAL_ERR, "write failed: " + tmp);
    }
    if (!write_all(fd, pack.data(), pack.size())) return bail(KF_INTERNAL_ERR, "write failed: " + tmp);
    if (fsync(fd) != 0) return bail(KF_INTERNAL_ERR, "fsync failed: " + tmp);
    close(fd);
    if (rename(tmp.c_str(), path.c_str()) != 0) {
        unlink(tmp.c_str());
        err = "rename to " + path + " failed";
        return KF_INTERNAL_ERR;
    }
    return KF_OK;
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
    if (const JSON* mt = cfg.get("model_type"))
        if (mt->kind == JSON::STR && mt->str != "qwen3")
            return bail(KF_INVALID_ARGS, "config.json: model_type '" + mt->str + "' (this path builds the Qwen3 decoder: RMSNorm, q/k-norm, rotate-half RoPE, SwiGLU, no projection biases)");
    {
        std::string why;
        if (!Fish::ShapeServed(card, why)) return bail(KF_UNSUPPORTED_DATATYPE, "config.json: " + why); /* before any tensor is quantised */
    }

    K_SafeTensors st;
    int rc = st.OpenDir(dir);
    if (rc != KF_OK) return bail(rc, st.err);
    for (const auto& t : st.tensors) { /* a Qwen2-style checkpoint would decode to wrong tokens with its biases dropped */
        const std::string& n = t.name;
        if (n.size() > 5 && n.compare(n.size() - 5, 5, ".bias") == 0 && (n.find("self_attn.") != std::string::npos || n.find("mlp.") != std::string::npos))
            return bail(KF_INVALID_ARGS, "tensor '" + n + "': projection biases are not part of this decoder (Qwen3 has none)");
    }

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

// ------------------------------------------------------------------------------------------------ Fish <-> .kun
namespace {
struct KunSlot {
    std::string name;
    hGTensor t;
};
// the parameters of the decode path under their Hugging Face names, in the order the neurons are built (embed, layers, final norm, head)
std::vector<KunSlot> kun_params(Fish* f) {
    std::vector<KunSlot> v;
    v.push_back({"model.embed_tokens.weight", f->embed.w});
    for (int l = 0; l < f->config.nLayer; l++) {
        const std::string p = "model.layers." + std::to_string(l) + ".";
        SelfAttention* a = f->attn[l].get();
        FFN* m = f->ffn[l].get();
        v.push_back({p + "input_layernorm.weight", a->norm.w});
        v.push_back({p + "self_attn.q_proj.weight", a->Q.w});
        v.push_back({p + "self_attn.k_proj.weight", a->K.w});
        v.push_back({p + "self_attn.v_proj.weight", a->V.w});
        if (a->normQ.w) v.push_back({p + "self_attn.q_norm.weight", a->normQ.w});
        if (a->normK.w) v.push_back({p + "self_attn.k_norm.weight", a->normK.w});
        v.push_back({p + "self_attn.o_proj.weight", a->proj_cat.w});
        v.push_back({p + "post_attention_layernorm.weight", m->norm.w});
        v.push_back({p + "mlp.gate_proj.weight", m->gate.w});
        v.push_back({p + "mlp.up_proj.weight", m->up.w});
        v.push_back({p + "mlp.down_proj.weight", m->down.w});
    }
    v.push_back({"model.norm.weight", f->final_norm.w});
    if (f->head.proj.w && f->head.proj.w != f->embed.w) v.push_back({"lm_head.weight", f->head.proj.w});  // a tied head "isRefer" and is skipped (Serialize.cpp:936)
    return v;
}
// "quantizer" section for one family of tensors (QUANT_CARD::Init, GeQuant.cpp:1231-1282: quant_method "RTN" -> RTN, "yyang" -> the ternary /
// binary forms, anything else -> RTNf for 4 bits and F8Ex for 8)
JSON quant_section(const GTensor& t) {
    JSON q = JSON::Object();
    const bool lowbit = t.type == typNUMBER::T_SIGN || t.type == typNUMBER::T_BINARY || t.type == typNUMBER::BOOL1;
    q["quant_method"] = JSON::Str(t.quant.isNormalFloat ? "NF" : lowbit ? "yyang" : t.type == typNUMBER::F8E5M2 ? "F8Ex" : "RTN");
    QuantCard qc;
    quant_range(t.type, false, qc);
    q["bits"] = JSON::Int(qc.bits);
    if (!t.quant.isNormalFloat && t.szGama) q["group_size"] = JSON::Int(t.quant.T_group);
    if (t.type == typNUMBER::Q4 && t.quant.qBias != 0) q["symmetric"] = JSON::Bool(true);  // ours: QUANT_CARD::isSymmetric is not a JSON key of the reference
    return q;
}
bool is_quant_type(typNUMBER t) { return t == typNUMBER::Q4 || t == typNUMBER::Q3 || t == typNUMBER::Q2 || t == typNUMBER::T_SIGN || t == typNUMBER::T_BINARY || t == typNUMBER::BOOL1; }
}  // namespace

// Fish -> `.kun` (the save branch of Fish::SAFETENSOR_Serialize, Serialize.cpp:911-963).  The config tensor carries what the reference's
// does -- {"vendor", "CLI_params": {"config": {...}}, "tokenizer", "tensors"} -- with the model card under the keys CLI_params reads back
// (CLI_params.cpp:368-432, cases/qwen3/qwen3_596M_q4.json) and two keys of ours under "parameter" (rope_theta, rms_norm_eps) that the
// reference takes from the Hugging Face card instead.
int SaveKun(Fish* f, const std::string& path, std::string& err) {
    const MODEL_CARD& c = f->config;
    std::vector<KunSlot> params = kun_params(f);
    KunWriter w;
    for (const auto& s : params) {
        if (!s.t || !s.t->data) {
            err = "tensor '" + s.name + "' is empty";  // "[ST_SERIALIZE] \"%s\" is empty!" (Serialize.cpp:634)
            return KF_INVALID_ARGS;
        }
        if (s.t->qZero || s.t->qScale) {
            err = "tensor '" + s.name + "': vendor AutoAWQ tensors (explicit zeros / scales) are not written to .kun";
            return KF_UNSUPPORTED_DATATYPE;
        }
        const char* dn = K_FLOATS_name((int)s.t->type);
        if (!dn) {
            err = "tensor '" + s.name + "': type has no K_FLOATS name";
            return KF_UNSUPPORTED_DATATYPE;
        }
        std::vector<int64_t> shape = {s.t->ne[0]};
        if (s.t->ne[1] > 1) shape.push_back(s.t->ne[1]);
        w.Register(s.name, dn, shape, s.t->szData, s.t->szGama);
    }
    JSON js = JSON::Object();
    js["vendor"] = JSON::Str("gruai");
    JSON& cfg = js["CLI_params"]["config"];
    cfg["version"] = JSON::Str("0.1.0");
    JSON& jq = cfg["quantizer"];
    jq = JSON::Object();
    const GTensor &tl = *f->attn[0]->Q.w, &tm = *f->ffn[0]->gate.w, &te = *f->embed.w;
    if (is_quant_type(tl.type) || tl.type == typNUMBER::F8E5M2) {
        if (tl.szGama && !tl.quant.isNormalFloat) jq["group_size"] = JSON::Int(tl.quant.T_group);
        jq["self_attn"] = quant_section(tl);
    }
    if (is_quant_type(tm.type) || tm.type == typNUMBER::F8E5M2) jq["mlp"] = quant_section(tm);
    if (is_quant_type(te.type) || te.type == typNUMBER::F8E5M2) jq["embed_tokens"] = quant_section(te);
    if (f->head.proj.w && f->head.proj.w != f->embed.w && (is_quant_type(f->head.proj.w->type) || f->head.proj.w->type == typNUMBER::F8E5M2))
        jq["lm_head"] = quant_section(*f->head.proj.w);
    JSON& jm = cfg["model"];
    jm["arch"] = JSON::Str("QWEN3");
    jm["vocab_size"] = JSON::Int(c.vocab);
    JSON& jp = jm["parameter"];
    jp["Layer"] = JSON::Int(c.nLayer);
    JSON& jt = jp["transformer"];
    jt["Ctx"] = JSON::Int(c.n_ctx), jt["Embed"] = JSON::Int(c.nEmbed), jt["Ffn"] = JSON::Int(c.n_ff);
    jt["Head"] = JSON::Int(c.n_head), jt["KVHead"] = JSON::Int(c.n_head_kv), jt["head_dim"] = JSON::Int(c.head_dim);
    jp["tie_word_embeddings"] = JSON::Bool(f->head.proj.w == f->embed.w);
    jp["max_pos_embeddings"] = JSON::Int(c.n_ctx);
    jp["rope_theta"] = JSON::Real(c.rope_theta);
    jp["rms_norm_eps"] = JSON::Real(c.rms_eps);
    js["tokenizer"]["tokens"] = JSON::Str("");
    kf_ctx* ctx = f->ctx;
    KF_TRY(kf_sync(ctx));
    return w.Save(path, js, [&](size_t i, void* dst, size_t n) { return kf_d2h(ctx, dst, params[i].t->data, n); }, err);  // SerialGamaData(toHost): data||gama in one copy
}

// `.kun` -> Fish (SAFETENSOR2Gensors + SerialGamaData H2D, Serialize.cpp:965, huTensor.cu:413-458): the blobs go to the device as they are,
// nothing is re-quantised.
Fish* LoadKun(const std::string& path, int device, void* stream, int max_seq, int* rc_out, std::string& err) {
    auto bail = [&](int rc, const std::string& what) -> Fish* {
        if (rc_out) *rc_out = rc;
        err = what;
        return nullptr;
    };
    K_SafeTensors st;
    int rc = st.OpenFile(path);
    if (rc != KF_OK) return bail(rc, st.err);
    JSON js;
    std::string jerr;
    if (!st.Config(js, jerr)) return bail(KF_INVALID_ARGS, path + ": " + jerr);
    const JSON* vendor = js.get("vendor");
    if (!vendor || vendor->kind != JSON::STR || vendor->str != "gruai") return bail(KF_INVALID_ARGS, path + ": config lacks \"vendor\": \"gruai\"");
    const JSON* cfg = js.path({"CLI_params", "config"});  // SAFETENSOR_Load_jconfig (Serialize.cpp:496-534)
    const JSON* jp = cfg ? cfg->path({"model", "parameter"}) : nullptr;
    const JSON* jt = jp ? jp->get("transformer") : nullptr;
    if (!jt) return bail(KF_INVALID_ARGS, path + ": config lacks CLI_params.config.model.parameter.transformer");
    MODEL_CARD card;
    card.nLayer = (int)jp->number_or("Layer", 1);
    card.nEmbed = (int)jt->number_or("Embed", 0), card.n_ff = (int)jt->number_or("Ffn", card.nEmbed * 4);
    card.n_head = (int)jt->number_or("Head", 0), card.n_head_kv = (int)jt->number_or("KVHead", card.n_head);
    card.head_dim = (int)jt->number_or("head_dim", card.n_head ? card.nEmbed / card.n_head : 0);
    const int ctx_len = (int)jt->number_or("Ctx", 0);
    card.tie_word_embeddings = jp->bool_or("tie_word_embeddings", true);
    card.rope_theta = (float)jp->number_or("rope_theta", 1e6);
    card.rms_eps = card.qk_eps = (float)jp->number_or("rms_norm_eps", 1e-6);
    const ST_Tensor* emb_t = st.Find("model.embed_tokens.weight");
    if (!emb_t || emb_t->shape.size() != 2) return bail(KF_INVALID_ARGS, path + ": model.embed_tokens.weight missing");
    card.vocab = (int)cfg->path({"model"})->number_or("vocab_size", (double)emb_t->shape[0]);
    card.n_ctx = max_seq > 0 ? max_seq : ctx_len;
    if (card.nEmbed <= 0 || card.nLayer <= 0 || card.n_head <= 0 || card.n_head_kv <= 0 || card.head_dim <= 0 || card.n_ff <= 0 || card.vocab <= 0 || card.n_ctx <= 0)
        return bail(KF_INVALID_ARGS, path + ": incomplete model card");
    const JSON* jq = cfg->get("quantizer");
    const int group_dflt = jq ? (int)jq->number_or("group_size", 128) : 128;

    std::unique_ptr<Fish> f(new Fish());
    rc = f->Build(card, device, stream);
    if (rc != KF_OK) return bail(rc, "Fish::Build failed");
    // one tensor: type from its K_FLOATS dtype, quant card from the quantizer section whose key its name contains (G_Has_, GeQuant.cpp:1231)
    auto load = [&](const std::string& name, int ne0, int ne1, bool required, hGTensor* out) -> int {
        const ST_Tensor* s = st.Find(name);
        if (!s) {
            if (!required) return KF_OK;
            err = "tensor '" + name + "' not in " + path;
            return KF_INVALID_ARGS;
        }
        const int tp = K_FLOATS_type(s->dtype);
        const bool shape_ok = (s->shape.size() == 2 && s->shape[0] == ne0 && s->shape[1] == ne1) || (s->shape.size() == 1 && ne1 == 1 && s->shape[0] == ne0);
        if (tp < 0 || !shape_ok) {
            err = "tensor '" + name + "': unexpected dtype '" + s->dtype + "' or shape";
            return KF_INVALID_ARGS;
        }
        auto t = std::make_shared<GTensor>();
        t->name = name, t->type = (typNUMBER)tp, t->ne[0] = ne0, t->ne[1] = ne1;
        t->szData = s->szData, t->szGama = s->szGama;
        if (t->szData + t->szGama == 0) t->szData = s->end - s->begin;
        const JSON* sec = nullptr;
        if (jq)
            for (const auto& kv : jq->obj)
                if (kv.second.kind == JSON::OBJ && !kv.first.empty() && kv.first[0] != '#' && name.find(kv.first) != std::string::npos) sec = &kv.second;
        quant_range(t->type, sec && sec->bool_or("symmetric", false), t->quant);
        t->quant.T_group = sec ? (int)sec->number_or("group_size", group_dflt) : group_dflt;
        if (sec && t->type == typNUMBER::Q4) {
            const JSON* m = sec->get("quant_method");
            const std::string ms = m && m->kind == JSON::STR ? m->str : "";
            t->quant.isNormalFloat = ms.find("RTN") == std::string::npos && ms.find("AWQ") == std::string::npos && ms.find("yyang") == std::string::npos &&
                                     ms.find("bitnet") == std::string::npos;  // GeQuant.cpp:1271-1279
        }
        const size_t n = (size_t)ne0 * ne1;
        size_t want_data = n * 2, want_gama = 0;
        if (is_quant_type(t->type)) {
            want_data = n * t->quant.bits / 8;
            if (t->quant.isNormalFloat) {
                t->quant.T_group = 0;
                want_gama = ((size_t)ne0 + ne1 + 16 * (size_t)ne0) * 2;
            } else {
                if (t->quant.T_group <= 0 || n % (size_t)t->quant.T_group) {
                    err = "tensor '" + name + "': group size " + std::to_string(t->quant.T_group);
                    return KF_QUANT_ERR;
                }
                want_gama = ((size_t)ne0 + ne1 + 2 * (n / t->quant.T_group)) * 2;
            }
        } else if (t->type == typNUMBER::F8E5M2) {
            want_data = n;
        } else if (t->type != typNUMBER::BF16) {
            err = "tensor '" + name + "': dtype '" + s->dtype + "' is not served by the decode path";
            return KF_UNSUPPORTED_DATATYPE;
        }
        if (t->szData != want_data || t->szGama != want_gama) {
            err = "tensor '" + name + "': szData/szGama " + std::to_string(t->szData) + "/" + std::to_string(t->szGama) + " do not fit its type, shape and quant card (" +
                  std::to_string(want_data) + "/" + std::to_string(want_gama) + ")";
            return KF_INVALID_ARGS;
        }
        KF_TRY(t->LoadBlob(f->ctx, st.Data(*s), t->szData + t->szGama));
        *out = t;
        return KF_OK;
    };
    auto linear = [&](const std::string& name, int n_out, int n_in, SLP* slot) -> int {
        hGTensor t;
        KF_TRY(load(name, n_out, n_in, true, &t));
        slot->w = t, slot->nOut = n_out, slot->nIn = n_in;
        return f->EnsureLinearScratch(t->desc(), f->prefill_chunk);
    };
    const int C = card.nEmbed, qd = card.n_head * card.head_dim, kvd = card.n_head_kv * card.head_dim;
    auto ok = [&](int r) { return (rc = r) == KF_OK; };
    if (!ok(load("model.embed_tokens.weight", card.vocab, C, true, &f->embed.w))) return bail(rc, err);
    if (st.Find("lm_head.weight")) {
        if (!ok(linear("lm_head.weight", card.vocab, C, &f->head.proj))) return bail(rc, err);
    } else {
        f->head.proj.w = f->embed.w, f->head.proj.nOut = card.vocab, f->head.proj.nIn = C;
    }
    if (!ok(load("model.norm.weight", C, 1, true, &f->final_norm.w))) return bail(rc, err);
    for (int l = 0; l < card.nLayer; l++) {
        const std::string p = "model.layers." + std::to_string(l) + ".";
        SelfAttention* a = f->attn[l].get();
        FFN* m = f->ffn[l].get();
        if (!ok(load(p + "input_layernorm.weight", C, 1, true, &a->norm.w)) || !ok(load(p + "post_attention_layernorm.weight", C, 1, true, &m->norm.w)) ||
            !ok(load(p + "self_attn.q_norm.weight", card.head_dim, 1, false, &a->normQ.w)) || !ok(load(p + "self_attn.k_norm.weight", card.head_dim, 1, false, &a->normK.w)) ||
            !ok(linear(p + "self_attn.q_proj.weight", qd, C, &a->Q)) || !ok(linear(p + "self_attn.k_proj.weight", kvd, C, &a->K)) ||
            !ok(linear(p + "self_attn.v_proj.weight", kvd, C, &a->V)) || !ok(linear(p + "self_attn.o_proj.weight", C, qd, &a->proj_cat)) ||
            !ok(linear(p + "mlp.gate_proj.weight", card.n_ff, C, &m->gate)) || !ok(linear(p + "mlp.up_proj.weight", card.n_ff, C, &m->up)) ||
            !ok(linear(p + "mlp.down_proj.weight", C, card.n_ff, &m->down)))
            return bail(rc, err);
    }
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
// ---- `.kun` (the reference's own checkpoint container)
// Fish -> file; 0 or a negative code with kfh_last_error() set
int kfh_save_kun(void* h, const char* path) {
    std::string err;
    const int rc = SaveKun(reinterpret_cast<Fish*>(h), path, err);
    if (rc != KF_OK) g_st_err = err;
    return rc;
}
void* kfh_load_kun(const char* path, int device, void* stream, int max_seq, int* rc) {
    std::string err;
    Fish* f = LoadKun(path, device, stream, max_seq, rc, err);
    if (!f) g_st_err = err;
    return f;
}
// host-only writer (tests, tools): n entries with K_FLOATS dtype names, shapes as 4 int64 per entry (0 = unused), host `data||gama` blobs, and
// the config as JSON text (packed to msgpack inside)
int kfh_kun_write(const char* path, int n, const char* const* names, const char* const* dtypes, const int64_t* shape4, const uint64_t* szData, const uint64_t* szGama,
                  const void* const* blobs, const char* config_json) {
    JSON js;
    std::string err;
    if (!JSON::Parse(config_json, strlen(config_json), js, err) || js.kind != JSON::OBJ) {
        g_st_err = "config: " + err;
        return KF_INVALID_ARGS;
    }
    KunWriter w;
    for (int i = 0; i < n; i++) {
        std::vector<int64_t> shape;
        for (int d = 0; d < 4 && shape4[4 * i + d] > 0; d++) shape.push_back(shape4[4 * i + d]);
        w.Register(names[i], dtypes[i], shape, szData[i], szGama[i]);
    }
    const int rc = w.Save(path, js, [&](size_t i, void* dst, size_t nb) { memcpy(dst, blobs[i], nb); return (int)KF_OK; }, err);
    if (rc != KF_OK) g_st_err = err;
    return rc;
}
// the config tensor of an opened `.kun` as JSON text; returns the length needed (copy truncated to cap - 1), or a negative code
int64_t kfh_st_config_json(void* h, char* out, int64_t cap) {
    JSON js;
    std::string err;
    if (!reinterpret_cast<K_SafeTensors*>(h)->Config(js, err)) {
        g_st_err = err;
        return KF_INVALID_ARGS;
    }
    const std::string text = js.Dump();
    if (out && cap > 0) snprintf(out, (size_t)cap, "%s", text.c_str());
    return (int64_t)text.size() + 1;
}
// szData / szGama of entry i (0 / 0 for Hugging Face files)
int kfh_st_blob_sizes(void* h, int i, uint64_t* szData, uint64_t* szGama) {
    auto* st = reinterpret_cast<K_SafeTensors*>(h);
    if (i < 0 || i >= (int)st->tensors.size()) return KF_INVALID_ARGS;
    *szData = st->tensors[i].szData, *szGama = st->tensors[i].szGama;
    return KF_OK;
}
// JSON text <-> msgpack round trips for the tests (python's msgpack is the independent checker)
int64_t kfh_json_to_msgpack(const char* text, uint8_t* out, int64_t cap) {
    JSON js;
    std::string err;
    if (!JSON::Parse(text, strlen(text), js, err)) {
        g_st_err = err;
        return KF_INVALID_ARGS;
    }
    std::vector<uint8_t> pack;
    js.ToMsgpack(pack);
    if (out && cap >= (int64_t)pack.size()) memcpy(out, pack.data(), pack.size());
    return (int64_t)pack.size();
}
int64_t kfh_msgpack_to_json(const uint8_t* pack, int64_t n, char* out, int64_t cap) {
    JSON js;
    std::string err;
    if (!JSON::FromMsgpack(pack, (size_t)n, js, err)) {
        g_st_err = err;
        return KF_INVALID_ARGS;
    }
    const std::string text = js.Dump();
    if (out && cap > 0) snprintf(out, (size_t)cap, "%s", text.c_str());
    return (int64_t)text.size() + 1;
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
