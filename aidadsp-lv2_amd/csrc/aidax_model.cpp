// aidax_model.cpp — AIDA-X json -> aidax_model (host only).
//
// Mirrors what the reference reads, key by key:
//   loadModelFromPath      rt-neural-generic/src/rt-neural-generic.cpp:963-1044
//   custom_model_creator   rt-neural-generic/src/model_variant.hpp:62-71 (pattern), :656-875
//   size lists             variant/generate_variant_hpp.py:3-6
// Weight layouts are the Keras ones RTNeural's json loader consumes:
//   lstm  [I][4H] | [H][4H] | [4H]      columns i|f|c|o
//   gru   [I][3H] | [H][3H] | [2][3H]   columns z|r|h
//   dense [H][1]  | [1]
//   conv1d [k][in][out] | [out], keys kernel_size / dilation   (extension)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "aidax_internal.h"
#include "json_min.h"

namespace aidax {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) { g_last_error = msg; return code; }

namespace {

struct ArchError : std::runtime_error { using std::runtime_error::runtime_error; };

void flatten(const Json& j, std::vector<float>& out, std::vector<size_t>& shape, size_t depth = 0)
{
    if (j.is_array()) {
        if (shape.size() <= depth) shape.push_back(j.size());
        else if (shape[depth] != j.size()) throw JsonError("json: ragged weight array");
        for (const auto& e : j.arr) flatten(e, out, shape, depth + 1);
    } else if (j.is_number()) {
        out.push_back(static_cast<float>(j.num));       // json doubles narrowed to fp32, as RTNeural<float> does
    } else {
        throw JsonError("json: non-numeric weight");
    }
}

std::vector<float> weights(const Json& layer, size_t idx, const std::vector<size_t>& want)
{
    const Json& w = layer.at("weights").at(idx);
    std::vector<float> out;
    std::vector<size_t> shape;
    flatten(w, out, shape);
    if (shape != want) {
        std::ostringstream os;
        os << "json: layer weight " << idx << " has shape [";
        for (size_t d : shape) os << d << ",";
        os << "] expected [";
        for (size_t d : want) os << d << ",";
        os << "]";
        throw JsonError(os.str());
    }
    return out;
}

int activation_id(const Json& layer)
{
    const Json& a = layer["activation"];
    if (!a.is_string() || a.str.empty() || a.str == "linear") return 0;
    if (a.str == "tanh") return 1;
    if (a.str == "relu") return 2;
    if (a.str == "sigmoid") return 3;
    throw ArchError("unsupported activation '" + a.str + "'");
}

int last_int(const Json& arr_or_num)
{
    if (arr_or_num.is_array()) return arr_or_num.back().as_int();
    return arr_or_num.as_int();
}

bool in_list(int v, std::initializer_list<int> l)
{
    for (int x : l) if (x == v) return true;
    return false;
}

void build(const Json& j, aidax_model& m)
{
    // ---- scalars, rt-neural-generic.cpp:977-1013
    m.input_size = j.at("in_shape").back().as_int();
    if (m.input_size > kMaxInputs) throw ArchError("Value for input_size not supported");
    if (m.input_size < 1) throw ArchError("input_size < 1");
    if (j["in_skip"].is_number()) {
        m.input_skip = j["in_skip"].as_int();
        if (m.input_skip > 1) throw ArchError("Values for in_skip > 1 are not supported");
        m.input_skip = m.input_skip != 0;                       // :1048
    }
    if (j["in_gain"].is_number()) m.input_gain = db_to_coeff(static_cast<float>(j["in_gain"].num));
    if (j["out_gain"].is_number()) m.output_gain = db_to_coeff(static_cast<float>(j["out_gain"].num));
    if (j["metadata"]["samplerate"].is_number()) m.samplerate = static_cast<float>(j["metadata"]["samplerate"].num);
    else if (j["samplerate"].is_number()) m.samplerate = static_cast<float>(j["samplerate"].num);
    else m.samplerate = 48000.0f;                               // a string "48000" lands here (:1012)

    // ---- layers
    const Json& layers = j.at("layers");
    if (!layers.is_array() || layers.size() < 2) throw ArchError("Unable to identify a known model architecture!");
    const std::string first = layers.at(0).at("type").as_string();
    int cur = m.input_size;
    for (size_t li = 0; li < layers.size(); ++li) {
        const Json& L = layers.at(li);
        const std::string type = L.at("type").as_string();
        Layer out;
        out.in_size = cur;
        out.out_size = L.at("shape").back().as_int();
        const size_t I = static_cast<size_t>(cur), O = static_cast<size_t>(out.out_size);
        if (out.out_size < 1) throw ArchError("layer with non-positive size");
        if (type == "lstm") {
            out.type = Layer::LSTM;
            out.w0 = weights(L, 0, {I, 4 * O});
            out.w1 = weights(L, 1, {O, 4 * O});
            out.w2 = weights(L, 2, {4 * O});
        } else if (type == "gru") {
            out.type = Layer::GRU;
            out.w0 = weights(L, 0, {I, 3 * O});
            out.w1 = weights(L, 1, {O, 3 * O});
            out.w2 = weights(L, 2, {2, 3 * O});
        } else if (type == "dense") {
            out.type = Layer::DENSE;
            out.activation = activation_id(L);
            out.w0 = weights(L, 0, {I, O});
            out.w1 = weights(L, 1, {O});
        } else if (type == "conv1d") {
            out.type = Layer::CONV1D;
            out.activation = activation_id(L);
            out.ksize = last_int(L.at("kernel_size"));
            out.dilation = last_int(L.at("dilation"));
            if (out.ksize < 1 || out.dilation < 1) throw ArchError("bad conv1d geometry");
            out.w0 = weights(L, 0, {static_cast<size_t>(out.ksize), I, O});
            out.w1 = weights(L, 1, {O});
        } else {
            throw ArchError("Unable to identify a known model architecture! (layer type '" + type + "')");
        }
        m.n_weights += out.w0.size() + out.w1.size() + out.w2.size();
        cur = out.out_size;
        m.layers.push_back(std::move(out));
    }

    // ---- architecture: [rnn x n][dense -> 1]  or  [conv1d x n][dense -> 1]
    const Layer& last = m.layers.back();
    if (last.type != Layer::DENSE || last.out_size != 1 || last.activation != 0)
        throw ArchError("Unable to identify a known model architecture! (last layer must be Dense(H,1))");
    const Layer::Type body = m.layers.front().type;
    if (body == Layer::DENSE) throw ArchError("Unable to identify a known model architecture!");
    m.hidden = m.layers.front().out_size;
    m.n_rnn = static_cast<int>(m.layers.size()) - 1;
    for (int l = 0; l < m.n_rnn; ++l)
        if (m.layers[l].type != body || m.layers[l].out_size != m.hidden)
            throw ArchError("Unable to identify a known model architecture! (mixed body layers)");
    m.cell = body == Layer::LSTM ? AIDAX_CELL_LSTM : body == Layer::GRU ? AIDAX_CELL_GRU : AIDAX_CELL_CONV;
    (void)first;
    m.in_reference_set = m.n_rnn == 1 && m.cell != AIDAX_CELL_CONV &&
                         in_list(m.hidden, {8, 12, 16, 20, 24, 32, 40, 64, 80});

    // ---- embedded golden vectors consumed by testModel (:1067-1073)
    if (j["input_batch"].is_array() && j["output_batch"].is_array()) {
        std::vector<size_t> sh;
        flatten(j["input_batch"], m.golden_in, sh);
        sh.clear();
        flatten(j["output_batch"], m.golden_out, sh);
        if (m.golden_in.size() != m.golden_out.size()) { m.golden_in.clear(); m.golden_out.clear(); }
    }
}

}  // namespace

bool model_supported(const aidax_model& m);   // aidax_pool.cpp

int load_from_text(const char* text, size_t len, const char* label, aidax_model** out)
{
    if (!text || !out) return fail(AIDAX_ERR_ARG, "null argument");
    *out = nullptr;
    auto m = std::make_unique<aidax_model>();
    m->path = label ? label : "";
    try {
        const Json j = Json::parse(text, len);
        build(j, *m);
    } catch (const ArchError& e) {
        return fail(AIDAX_ERR_ARCH, std::string("Error loading model: ") + e.what());
    } catch (const std::exception& e) {
        return fail(AIDAX_ERR_JSON, std::string("Unable to load json file: ") + m->path + "\nError: " + e.what());
    }
    if (!model_supported(*m))
        return fail(AIDAX_ERR_ARCH, "Error loading model: Unable to identify a known model architecture! (no kernel for this cell/hidden size)");
    // AIDAX_STRICT_REFERENCE_SET=1: accept exactly what custom_model_creator accepts (model_variant.hpp:6-59,
    // rt-neural-generic.cpp:1025-1026) and nothing of this build's extensions
    const char* strict = std::getenv("AIDAX_STRICT_REFERENCE_SET");
    if (strict && strict[0] == '1' && !m->in_reference_set)
        return fail(AIDAX_ERR_ARCH, "Error loading model: Unable to identify a known model architecture!");
    *out = m.release();
    return AIDAX_OK;
}

}  // namespace aidax

using namespace aidax;

extern "C" {

AIDAX_API const char* aidax_last_error(void) { return g_last_error.c_str(); }
AIDAX_API const char* aidax_version(void) { return "aidax-mi355x 0.1 (gfx950)"; }

AIDAX_API int aidax_model_load(const char* json_path, aidax_model** out)
{
    if (!json_path || !out) return fail(AIDAX_ERR_ARG, "null argument");
    *out = nullptr;
    std::ifstream f(json_path, std::ifstream::binary);
    if (!f) return fail(AIDAX_ERR_IO, std::string("Unable to load json file: ") + json_path);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    return load_from_text(text.data(), text.size(), json_path, out);
}

AIDAX_API int aidax_model_load_memory(const char* json_text, size_t len, const char* label, aidax_model** out)
{
    return load_from_text(json_text, len, label, out);
}

AIDAX_API int aidax_model_info(const aidax_model* m, aidax_model_info_t* info)
{
    if (!m || !info) return fail(AIDAX_ERR_ARG, "null argument");
    info->cell = m->cell;
    info->hidden = m->hidden;
    info->input_size = m->input_size;
    info->n_rnn_layers = m->n_rnn;
    info->input_skip = m->input_skip;
    info->input_gain = m->input_gain;
    info->output_gain = m->output_gain;
    info->samplerate = m->samplerate;
    info->n_golden = static_cast<int32_t>(m->golden_in.size());
    info->in_reference_set = m->in_reference_set ? 1 : 0;
    info->n_weights = m->n_weights;
    return AIDAX_OK;
}

AIDAX_API const char* aidax_model_path(const aidax_model* m) { return m ? m->path.c_str() : ""; }

AIDAX_API int aidax_model_golden(const aidax_model* m, float* in, float* out, uint32_t cap)
{
    if (!m) return fail(AIDAX_ERR_ARG, "null argument");
    const size_t n = m->golden_in.size() < cap ? m->golden_in.size() : cap;
    if (in && n) std::memcpy(in, m->golden_in.data(), n * sizeof(float));       // (a model without goldens: data() may be null)
    if (out && n) std::memcpy(out, m->golden_out.data(), n * sizeof(float));
    return static_cast<int>(n);
}

AIDAX_API void aidax_model_free(aidax_model* m) { delete m; }

}  // extern "C"
