// kf_safetensors.hpp -- reading Hugging Face checkpoints (config.json + *.safetensors) into the host-side Fish.
//
// Reference: K_SafeTensors / Fish::SAFETENSOR_Serialize / SAFETENSOR2Gensors (src/Manifold/Serialize.cpp:849-976), the HF card in
// CLI_params (src/Utils/CLI_params.cpp:2177-2300: hidden_size, num_hidden_layers, ..., rope_theta, tie_word_embeddings) and the
// quantise-on-load of GeQuant (GeQuant.cpp:144-200 -> RTN_x).  Only the read side, and only what the forward path needs: the file is
// mapped, the JSON header parsed (8-byte little-endian length + JSON, https://github.com/huggingface/safetensors format), tensors are
// copied to the device and quantised there by kf_quantize.  Plain C++17, no third-party JSON library.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace koifish {

// a parsed JSON value (just enough for safetensors headers, config.json and model.safetensors.index.json)
struct JSON {
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<JSON> arr;
    std::vector<std::pair<std::string, JSON>> obj;  // insertion order kept
    const JSON* get(const std::string& key) const;
    double number_or(const std::string& key, double dflt) const;
    bool bool_or(const std::string& key, bool dflt) const;
    // returns false on malformed input; `err` gets a short description with the byte offset
    static bool Parse(const char* text, size_t n, JSON& out, std::string& err);
};

struct ST_Tensor {
    std::string name, dtype;  // "BF16", "F16", "F32", "I32", ...
    std::vector<int64_t> shape;
    size_t begin = 0, end = 0;  // data_offsets relative to the byte buffer that follows the header
    int file = 0;
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
};

}  // namespace koifish
