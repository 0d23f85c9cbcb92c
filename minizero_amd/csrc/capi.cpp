// extern "C" surface of libmzgpu for the network and the search pool (include/mzgpu.h).
// The worker and environment entry points live in worker.cpp / env.cpp.
#include "net.h"
#include "pool.h"
#include <atomic>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace mz {
static thread_local char g_err[1024] = "";
void setError(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* lastError() { return g_err; }

static std::atomic<uint64_t> g_weight_file_reads{0};
uint64_t weightFileReads() { return g_weight_file_reads.load(); }

bool readWeightFile(const std::string& path_in, mz_net_desc* desc, std::vector<float>* weights)
{
    g_weight_file_reads.fetch_add(1);
    std::string path = path_in;
    mz_net_desc d;
    std::vector<float> w;
    const bool is_pt = path.size() > 3 && path.compare(path.size() - 3, 3, ".pt") == 0;
    FILE* probe = is_pt ? fopen(path.c_str(), "rb") : nullptr;
    if (probe) {
        fclose(probe);
        std::string err;
        if (!readTorchScript(path, &d, &w, &err)) { setError("%s", err.c_str()); return false; }
    } else {
        if (is_pt) { path = path.substr(0, path.size() - 3) + ".mzw"; }
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) { setError("cannot open %s%s", path_in.c_str(), is_pt ? (" (nor " + path + ")").c_str() : ""); return false; }
        char magic[4];
        uint64_t count = 0;
        bool ok = fread(magic, 1, 4, f) == 4 && memcmp(magic, "MZW1", 4) == 0 && fread(&d, sizeof(d), 1, f) == 1 && fread(&count, sizeof(count), 1, f) == 1 &&
                  count < (1ull << 32);
        if (ok) {
            w.resize(count);
            ok = fread(w.data(), sizeof(float), count, f) == count;
        }
        fclose(f);
        if (!ok) { setError("bad weight file %s", path.c_str()); return false; }
        d.game_name[sizeof(d.game_name) - 1] = 0;
    }
    if (!netValidateDesc(d)) { return false; }
    if (static_cast<long>(w.size()) != netParamCount(d)) {
        setError("%s: %zu floating-point values in the file, the %d-block network of its hyper-parameters has %ld", path.c_str(), w.size(), d.num_blocks,
                 netParamCount(d));
        return false;
    }
    if (desc) { *desc = d; }
    if (weights) { weights->swap(w); }
    return true;
}
} // namespace mz

struct mz_net { mz::Net net; };
struct mz_pool { mz::Pool pool; };

extern "C" {

const char* mz_last_error(void) { return mz::lastError(); }

int mz_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

long mz_net_param_count(const mz_net_desc* desc)
{
    if (!desc || !mz::netValidateDesc(*desc)) { return MZ_ERR_ARG; }
    return mz::netParamCount(*desc);
}

int mz_net_generate_weights(const mz_net_desc* desc, uint64_t seed, float* out)
{
    if (!desc || !out || !mz::netValidateDesc(*desc)) { return MZ_ERR_ARG; }
    return mz::netGenerate(*desc, seed, out) ? MZ_OK : MZ_ERR_ARG;
}

int mz_net_read_pt(const char* path, mz_net_desc* desc_out, float* weights_out, size_t capacity, size_t* count_out)
{
    if (!path) { mz::setError("mz_net_read_pt: NULL path"); return MZ_ERR_ARG; }
    mz_net_desc d;
    std::vector<float> w;
    std::string err;
    if (!mz::readTorchScript(path, &d, &w, &err)) { mz::setError("%s", err.c_str()); return MZ_ERR_ARG; }
    if (!mz::netValidateDesc(d)) { return MZ_ERR_ARG; }
    if (static_cast<long>(w.size()) != mz::netParamCount(d)) {
        mz::setError("%s: %zu floating-point values in the archive, the %d-block network of its hyper-parameters has %ld", path, w.size(), d.num_blocks,
                     mz::netParamCount(d));
        return MZ_ERR_ARG;
    }
    if (desc_out) { *desc_out = d; }
    if (count_out) { *count_out = w.size(); }
    if (weights_out) {
        if (capacity < w.size()) { mz::setError("mz_net_read_pt: buffer of %zu floats, %zu needed", capacity, w.size()); return MZ_ERR_ARG; }
        memcpy(weights_out, w.data(), w.size() * sizeof(float));
    }
    return MZ_OK;
}

int mz_net_read_weight_file(const char* path, mz_net_desc* desc_out, float* weights_out, size_t capacity, size_t* count_out)
{
    if (!path) { mz::setError("mz_net_read_weight_file: NULL path"); return MZ_ERR_ARG; }
    mz_net_desc d;
    std::vector<float> w;
    if (!mz::readWeightFile(path, &d, &w)) { return MZ_ERR_ARG; }
    if (desc_out) { *desc_out = d; }
    if (count_out) { *count_out = w.size(); }
    if (weights_out) {
        if (capacity < w.size()) { mz::setError("mz_net_read_weight_file: buffer of %zu floats, %zu needed", capacity, w.size()); return MZ_ERR_ARG; }
        memcpy(weights_out, w.data(), w.size() * sizeof(float));
    }
    return MZ_OK;
}

// one open + one parse, kept: what a driver of several devices hands to each of them (include/minizero/actor_group.h)
struct mz_weights { mz_net_desc desc; std::vector<float> data; };
mz_weights* mz_weights_read(const char* path)
{
    if (!path) { mz::setError("mz_weights_read: NULL path"); return nullptr; }
    std::unique_ptr<mz_weights> w(new mz_weights());
    if (!mz::readWeightFile(path, &w->desc, &w->data)) { return nullptr; }
    return w.release();
}
const mz_net_desc* mz_weights_desc(const mz_weights* w) { return w ? &w->desc : nullptr; }
const float* mz_weights_data(const mz_weights* w) { return w ? w->data.data() : nullptr; }
size_t mz_weights_count(const mz_weights* w) { return w ? w->data.size() : 0; }
void mz_weights_free(mz_weights* w) { delete w; }
uint64_t mz_weight_file_reads(void) { return mz::weightFileReads(); }

mz_net* mz_net_create(int device, const mz_net_desc* desc, const float* weights, size_t count)
{
    if (!desc || !weights) { mz::setError("mz_net_create: NULL argument"); return nullptr; }
    std::unique_ptr<mz_net> n(new mz_net());
    if (n->net.init(device, *desc, weights, count) != MZ_OK) { return nullptr; }
    return n.release();
}
int mz_net_reload(mz_net* net, const float* weights, size_t count)
{
    if (!net || !weights) { mz::setError("mz_net_reload: NULL argument"); return MZ_ERR_ARG; }
    return net->net.reload(weights, count);
}
void mz_net_destroy(mz_net* net) { delete net; }
int mz_net_set_precision(mz_net* net, int mode)
{
    if (!net) { mz::setError("NULL network"); return MZ_ERR_ARG; }
    return net->net.setPrecision(mode);
}
int mz_net_get_desc(const mz_net* net, mz_net_desc* out)
{
    if (!net || !out) { return MZ_ERR_ARG; }
    *out = net->net.desc_;
    return MZ_OK;
}
int mz_net_forward_az(mz_net* net, const float* features, int batch, float* policy, float* policy_logit, float* value, int where)
{
    if (!net) { mz::setError("NULL network"); return MZ_ERR_ARG; }
    return net->net.forwardAZ_any(features, batch, policy, policy_logit, value, where);
}
int mz_net_initial(mz_net* net, const float* features, int batch, float* policy, float* policy_logit, float* value, float* hidden_state, int where)
{
    if (!net) { mz::setError("NULL network"); return MZ_ERR_ARG; }
    return net->net.initial_any(features, batch, policy, policy_logit, value, hidden_state, where);
}
int mz_net_recurrent(mz_net* net, const float* hidden_in, const float* action_plane, int batch, float* policy, float* policy_logit, float* value,
                     float* reward, float* hidden_out, int where)
{
    if (!net) { mz::setError("NULL network"); return MZ_ERR_ARG; }
    return net->net.recurrent_any(hidden_in, action_plane, batch, policy, policy_logit, value, reward, hidden_out, where);
}
int mz_net_time_forward(mz_net* net, int batch, int iters, float* ms_total, float* ms_conv3x3, double* conv_flops_per_forward)
{
    if (!net) { mz::setError("NULL network"); return MZ_ERR_ARG; }
    return net->net.timeForward(batch, iters, ms_total, ms_conv3x3, conv_flops_per_forward);
}

int mz_net_time_tower_conv(mz_net* net, int batch, int iters, float* ms_per_launch, double* flops_per_launch, double* bytes_per_launch)
{
    if (!net) { mz::setError("NULL network"); return MZ_ERR_ARG; }
    return net->net.timeTowerConv(batch, iters, ms_per_launch, flops_per_launch, bytes_per_launch);
}

float mz_invert_value(float v) { return mz::invertValueHost(v); }

mz_pool* mz_pool_create(int device, int games, int nodes_per_game, int action_size, const mz_search_cfg* cfg)
{
    if (!cfg) { mz::setError("mz_pool_create: NULL cfg"); return nullptr; }
    std::unique_ptr<mz_pool> p(new mz_pool());
    if (p->pool.init(device, games, nodes_per_game, action_size, *cfg, nullptr) != MZ_OK) { return nullptr; }
    return p.release();
}
void mz_pool_destroy(mz_pool* pool) { delete pool; }
int mz_pool_reset_search(mz_pool* pool, const uint8_t* mask, const int* root_player)
{
    if (!pool) { mz::setError("NULL pool"); return MZ_ERR_ARG; }
    return pool->pool.resetSearch(mask, root_player);
}
int mz_pool_select(mz_pool* pool, const int* start_node, int* path_len, int* paths, int* path_action)
{
    if (!pool) { mz::setError("NULL pool"); return MZ_ERR_ARG; }
    return pool->pool.select(start_node, path_len, paths, path_action);
}
int mz_pool_max_depth(const mz_pool* pool) { return pool ? pool->pool.v_.max_depth : MZ_ERR_ARG; }
int mz_pool_expand_backup(mz_pool* pool, const int* cand_count, const int* cand_action, const float* cand_policy, const float* cand_logit,
                          const int* cand_player, const float* value, const float* reward)
{
    if (!pool) { mz::setError("NULL pool"); return MZ_ERR_ARG; }
    return pool->pool.expandBackup(cand_count, cand_action, cand_policy, cand_logit, cand_player, value, reward);
}
int mz_pool_root_set_noise(mz_pool* pool, const uint8_t* mask, const float* policy, const float* logit, const float* noise)
{
    if (!pool) { mz::setError("NULL pool"); return MZ_ERR_ARG; }
    return pool->pool.rootSetNoise(mask, policy, logit, noise);
}
int mz_pool_root_read(mz_pool* pool, int* num_children, int* action, float* count, float* mean, float* policy, float* logit, float* noise,
                      float* value, float* reward, float* root_count, float* root_mean, float* root_value, float* bound_lo, float* bound_hi,
                      int* bound_size)
{
    if (!pool) { mz::setError("NULL pool"); return MZ_ERR_ARG; }
    return pool->pool.rootRead(num_children, action, count, mean, policy, logit, noise, value, reward, root_count, root_mean, root_value, bound_lo,
                               bound_hi, bound_size);
}
int mz_pool_read_nodes(mz_pool* pool, int game, int n, int* action, int* player, int* num_children, int* first_child, float* mean, float* count,
                       float* policy, float* logit, float* noise, float* value, float* reward)
{
    if (!pool) { mz::setError("NULL pool"); return MZ_ERR_ARG; }
    return pool->pool.readNodes(game, n, action, player, num_children, first_child, mean, count, policy, logit, noise, value, reward);
}
int mz_pool_num_nodes(mz_pool* pool, int game)
{
    if (!pool) { mz::setError("NULL pool"); return MZ_ERR_ARG; }
    return pool->pool.numNodes(game);
}

} // extern "C"
