// kf_safetensors.hpp -- reading Hugging Face checkpoints (config.json + *.safetensors) into the host-side Fish, and reading / writing the
// reference's own `.kun` checkpoints (a safetensors container whose tensors are the `data||gama` blobs, plus a msgpack config tensor).
//
// Reference: K_SafeTensors / Fish::SAFETENSOR_Serialize / SAFETENSOR2Gensors (src/Manifold/Serialize.cpp:849-976), the HF card in
// CLI_params (src/Utils/CLI_params.cpp:2177-2300: hidden_size, num_hidden_layers, ..., rope_theta, tie_word_embeddings) and the
// quantise-on-load of GeQuant (GeQuant.cpp:144-200 -> RTN_x).  Only the read side, and only what the forward path needs: the file is
// mapped, the JSON header parsed (8-byte little-endian length + JSON, https://github.com/huggingface/safetensors format), tensors are
// copied to the device and quantised there by kf_quantize.  Plain C++17, no third-party JSON library.
#pragma once
#include <cstdint>
#include <functional>
#include <initializer_list>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace koifish {

// a parsed JSON value (just enough for safetensors headers, config.json and model.safetensors.index.json)
struct JSON {
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
    bool b = false;
    bool is_int = false;  // the token had no fraction / exponent (msgpack writes it as an integer)
    double num = 0.0;
    std::string str;
    std::vector<JSON> arr;
    std::vector<std::pair<std::string, JSON>> obj;  // insertion order kept
    const JSON* get(const std::string& key) const;
    double number_or(const std::string& key, double dflt) const;
    bool bool_or(const std::string& key, bool dflt) const;
    // returns false on malformed input; `err` gets a short description with the byte offset
    static bool Parse(const char* text, size_t n, JSON& out, std::string& err);
    // builders (insertion order kept, like the reference's nlohmann::ordered_json, src/Utils/json.hpp:24571)
    static JSON Str(const std::string& s);
    static JSON Int(int64_t v);
    static JSON Real(double v);
    static JSON Bool(bool v);
    static JSON Object();
    static JSON Array();
    JSON& operator[](const std::string& key);  // NUL -> OBJ on first use; appends a missing key
    const JSON* path(std::initializer_list<const char*> keys) const;
    std::string Dump() const;  // compact text, the form ordered_json::dump() gives ("{\"a\":1,\"b\":[1,2]}")
    // MessagePack in the encoding nlohmann's to_msgpack chooses (smallest integer form, float32 when exact): K_SafeTensors::insertJS
    // stores the config this way (src/Tensor/Safetensors.hpp:87-102)
    void ToMsgpack(std::vector<uint8_t>& out) const;
    static bool FromMsgpack(const uint8_t* p, size_t n, JSON& out, std::string& err);
};

struct ST_Tensor {
    std::string name, dtype;  // "BF16", "F16", "F32", "I32", ...
    std::vector<int64_t> shape;
    size_t begin = 0, end = 0;  // data_offsets relative to the byte buffer that follows the header
    int file = 0;
    size_t szData = 0, szGama = 0;  // `.kun` entries only (GTensor::jDesc, Serialize.cpp:61-103); 0 in Hugging Face files
};

// One or several mapped .safetensors files (a sharded checkpoint has model.safetensors.index.json naming them)
struct K_SafeTensors {
    struct File {
        std::string path;
        int fd = -1;
        void* map = nullptr;
        size_t size = 0, data_base = 0;
    };
    std::vector<File> files;
    std::vector<ST_Tensor> tensors;
    std::map<std::string, int> index;
    std::map<std::string, std::string> metadata;
    std::string err;

    ~K_SafeTensors();
    int OpenFile(const std::string& path);  // KF_OK or a negative KOIFISH code; err has the reason
    int OpenDir(const std::string& dir);    // model.safetensors, or every shard of model.safetensors.index.json
    const ST_Tensor* Find(const std::string& name) const;
    const void* Data(const ST_Tensor& t) const;
    // `.kun`: the msgpack config tensor "__koifish__config__" decoded (K_SafeTensors::initJS, Safetensors.hpp:121-146); false if the file has none
    bool Config(JSON& out, std::string& err) const;
    static const char* config_key() { return "__koifish__config__"; }  // src/Tensor/Safetensors.cpp:13
};

// Writer of a `.kun` file (K_SafeTensors::Register / insertJS / _to_ofs, Serialize.cpp:849-871, 554-665): entries are registered in order, each
// one a `data||gama` blob of szData + szGama bytes; Save writes [u64 header length][JSON header][blobs...][msgpack config] to a temporary
// file in the same directory and renames it over `path`.  `fetch` is called once per entry, in order, to fill `dst` (host memory).
struct KunWriter {
    struct Entry {
        std::string name, dtype;  // dtype: K_FLOATS name ("Q<4>", "TERNARY", "BINARY", "BF16(E8)", "F8E5M2", ...; src/g_float.hpp:127-151)
        std::vector<int64_t> shape;
        size_t szData = 0, szGama = 0, begin = 0;
    };
    std::vector<Entry> entries;
    size_t offset = 0;
    size_t Register(const std::string& name, const std::string& dtype, const std::vector<int64_t>& shape, size_t szData, size_t szGama);
    // jsConfig gets ["tensors"][name] = offset for every entry (Serialize.cpp:938) before it is packed
    int Save(const std::string& path, JSON jsConfig, const std::function<int(size_t i, void* dst, size_t nbytes)>& fetch, std::string& err);
};

const char* K_FLOATS_name(int typ);              // typNUMBER -> the reference's K_FLOATS name; nullptr when the type has none
int K_FLOATS_type(const std::string& name);      // the inverse (also accepts the aliases "F32", "F16", "BF16"); -1 when unknown

}  // namespace koifish
