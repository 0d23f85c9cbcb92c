// minizero::network facade over libmzgpu — same class names, method names, argument meaning and ownership
// as the reference (ref network/network.h:10-56, alphazero_network.h:13-121, muzero_network.h:14-190,
// create_network.h:11-30), no LibTorch.  Header-only; link with -lmzgpu.
//
// Differences a maintainer must know:
//   * loadModel("x.pt", gpu) reads the TorchScript archive the reference's trainer writes natively (mz_net_read_weight_file: zip +
//     pickle, no LibTorch); if "x.pt" does not exist it opens the sibling "x.mzw" (flat blob of minizero_amd/export_weights.py: magic,
//     mz_net_desc, count, f32 data) and keeps "x.pt" as getNetworkFileName()
//   * gpu_id == -1 (CPU) is not supported: loadModel prints the library error and aborts like the reference's
//     c10::Error path (network.cpp:20-26)
#pragma once
#include "../mzgpu.h"
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

namespace minizero::network {

class NetworkOutput {
public:
    virtual ~NetworkOutput() = default;
};

inline bool readWeightFile(const std::string& nn_file_name, mz_net_desc& desc, std::vector<float>& weights)
{
    size_t count = 0;
    if (mz_net_read_weight_file(nn_file_name.c_str(), &desc, nullptr, 0, &count) != MZ_OK) { std::cerr << mz_last_error() << std::endl; return false; }
    weights.resize(count);
    if (mz_net_read_weight_file(nn_file_name.c_str(), &desc, weights.data(), count, &count) != MZ_OK) { std::cerr << mz_last_error() << std::endl; return false; }
    return true;
}

class Network {
public:
    Network() { desc_ = mz_net_desc(); gpu_id_ = -1; }
    virtual ~Network() { if (net_ && own_) { mz_net_destroy(net_); } }

    virtual void loadModel(const std::string& nn_file_name, const int gpu_id)
    {
        gpu_id_ = gpu_id;
        network_file_name_ = nn_file_name;
        std::vector<float> w;
        if (!readWeightFile(nn_file_name, desc_, w)) { assert(false); std::abort(); }
        if (net_) {
            if (mz_net_reload(net_, w.data(), w.size()) != MZ_OK) { std::cerr << mz_last_error() << std::endl; assert(false); std::abort(); }
        } else {
            net_ = mz_net_create(gpu_id, &desc_, w.data(), w.size());
            if (!net_) { std::cerr << mz_last_error() << std::endl; assert(false); std::abort(); }
        }
    }
    virtual std::string toString() const
    {
        std::ostringstream oss;
        oss << "GPU ID: " << gpu_id_ << std::endl
            << "Number of input channels: " << getNumInputChannels() << std::endl
            << "Input channel height: " << getInputChannelHeight() << std::endl
            << "Input channel width: " << getInputChannelWidth() << std::endl
            << "Number of hidden channels: " << getNumHiddenChannels() << std::endl
            << "Hidden channel height: " << getHiddenChannelHeight() << std::endl
            << "Hidden channel width: " << getHiddenChannelWidth() << std::endl
            << "Number of blocks: " << getNumBlocks() << std::endl
            << "Action size: " << getActionSize() << std::endl
            << "Number of value hidden channels: " << getNumValueHiddenChannels() << std::endl
            << "Discrete value size: " << getDiscreteValueSize() << std::endl
            << "Game name: " << getGameName() << std::endl
            << "Network type name: " << getNetworkTypeName() << std::endl
            << "Network file name: " << getNetworkFileName() << std::endl;
        return oss.str();
    }

    inline int getGPUID() const { return gpu_id_; }
    inline int getNumInputChannels() const { return desc_.num_input_channels; }
    inline int getInputChannelHeight() const { return desc_.input_channel_height; }
    inline int getInputChannelWidth() const { return desc_.input_channel_width; }
    inline int getNumHiddenChannels() const { return desc_.num_hidden_channels; }
    inline int getHiddenChannelHeight() const { return desc_.hidden_channel_height; }
    inline int getHiddenChannelWidth() const { return desc_.hidden_channel_width; }
    inline int getNumBlocks() const { return desc_.num_blocks; }
    inline int getActionSize() const { return desc_.action_size; }
    inline int getNumValueHiddenChannels() const { return desc_.num_value_hidden_channels; }
    inline int getDiscreteValueSize() const { return desc_.discrete_value_size; }
    inline std::string getGameName() const { return desc_.game_name; }
    inline std::string getNetworkTypeName() const { return desc_.type == 0 ? "alphazero" : (desc_.type == 1 ? "muzero" : "muzero_atari"); }
    inline std::string getNetworkFileName() const { return network_file_name_; }
    inline mz_net* handle() const { return net_; }

protected:
    int gpu_id_;
    mz_net_desc desc_;
    std::string network_file_name_;
    mz_net* net_ = nullptr;
    bool own_ = true;
};

class AlphaZeroNetworkOutput : public NetworkOutput {
public:
    float value_;
    std::vector<float> policy_;
    std::vector<float> policy_logits_;
    explicit AlphaZeroNetworkOutput(int policy_size) : value_(0.0f), policy_(policy_size, 0.0f), policy_logits_(policy_size, 0.0f) {}
};

class AlphaZeroNetwork : public Network {
public:
    AlphaZeroNetwork() { clear(); }
    void loadModel(const std::string& nn_file_name, const int gpu_id) override
    {
        assert(batch_size_ == 0);
        Network::loadModel(nn_file_name, gpu_id);
        clear();
    }
    // thread-safe like the reference (alphazero_network.h:48-61): index under a mutex, copy outside it
    int pushBack(std::vector<float> features)
    {
        const size_t fs = static_cast<size_t>(getNumInputChannels()) * getInputChannelHeight() * getInputChannelWidth();
        assert(features.size() == fs && batch_size_ < kReserved_batch_size);
        int index;
        {
            std::lock_guard<std::mutex> lock(mutex_);
            index = batch_size_++;
        }
        std::copy(features.begin(), features.end(), input_.begin() + static_cast<size_t>(index) * fs);
        return index;
    }
    std::vector<std::shared_ptr<NetworkOutput>> forward()
    {
        assert(batch_size_ > 0);
        const int A = getActionSize();
        std::vector<float> policy(static_cast<size_t>(batch_size_) * A), logit(policy.size()), value(batch_size_);
        if (mz_net_forward_az(net_, input_.data(), batch_size_, policy.data(), logit.data(), value.data(), MZ_HOST) != MZ_OK) {
            std::cerr << mz_last_error() << std::endl;
            assert(false);
            std::abort();
        }
        std::vector<std::shared_ptr<NetworkOutput>> outs;
        for (int i = 0; i < batch_size_; ++i) {
            auto o = std::make_shared<AlphaZeroNetworkOutput>(A);
            std::copy(policy.begin() + static_cast<size_t>(i) * A, policy.begin() + static_cast<size_t>(i + 1) * A, o->policy_.begin());
            std::copy(logit.begin() + static_cast<size_t>(i) * A, logit.begin() + static_cast<size_t>(i + 1) * A, o->policy_logits_.begin());
            o->value_ = value[i];
            outs.emplace_back(o);
        }
        clear();
        return outs;
    }
    inline int getBatchSize() const { return batch_size_; }

protected:
    inline void clear()
    {
        batch_size_ = 0;
        const size_t fs = static_cast<size_t>(std::max(0, getNumInputChannels() * getInputChannelHeight() * getInputChannelWidth()));
        input_.resize(fs * kReserved_batch_size);
    }
    int batch_size_ = 0;
    std::mutex mutex_;
    std::vector<float> input_;
    const int kReserved_batch_size = 4096;
};

class MuZeroNetworkOutput : public NetworkOutput {
public:
    float value_, reward_;
    std::vector<float> policy_, policy_logits_, hidden_state_;
    MuZeroNetworkOutput(int policy_size, int hidden_state_size)
        : value_(0.0f), reward_(0.0f), policy_(policy_size, 0.0f), policy_logits_(policy_size, 0.0f), hidden_state_(hidden_state_size, 0.0f) {}
};

class MuZeroNetwork : public Network {
public:
    void loadModel(const std::string& nn_file_name, const int gpu_id) override
    {
        Network::loadModel(nn_file_name, gpu_id);
        initial_.clear();
        rec_hidden_.clear();
        rec_action_.clear();
    }
    std::string toString() const override
    {
        std::ostringstream oss;
        oss << Network::toString() << "Number of action feature channels: " << getNumActionFeatureChannels() << std::endl;
        return oss.str();
    }
    int pushBackInitialData(std::vector<float> features)
    {
        std::lock_guard<std::mutex> lock(initial_mutex_);
        initial_.insert(initial_.end(), features.begin(), features.end());
        return getInitialInputBatchSize() - 1;
    }
    int pushBackRecurrentData(std::vector<float> features, std::vector<float> actions)
    {
        std::lock_guard<std::mutex> lock(recurrent_mutex_);
        rec_hidden_.insert(rec_hidden_.end(), features.begin(), features.end());
        rec_action_.insert(rec_action_.end(), actions.begin(), actions.end());
        return getRecurrentInputBatchSize() - 1;
    }
    inline std::vector<std::shared_ptr<NetworkOutput>> initialInference()
    {
        const int B = getInitialInputBatchSize();
        assert(B > 0);
        Buffers b(B, getActionSize(), hiddenSize());
        check(mz_net_initial(net_, initial_.data(), B, b.policy.data(), b.logit.data(), b.value.data(), b.hidden.data(), MZ_HOST));
        initial_.clear();
        return b.outputs();
    }
    inline std::vector<std::shared_ptr<NetworkOutput>> recurrentInference()
    {
        const int B = getRecurrentInputBatchSize();
        assert(B > 0);
        Buffers b(B, getActionSize(), hiddenSize());
        check(mz_net_recurrent(net_, rec_hidden_.data(), rec_action_.data(), B, b.policy.data(), b.logit.data(), b.value.data(), b.reward.data(),
                               b.hidden.data(), MZ_HOST));
        rec_hidden_.clear();
        rec_action_.clear();
        return b.outputs();
    }
    inline int getNumActionFeatureChannels() const { return desc_.num_action_feature_channels; }
    inline int getInitialInputBatchSize() const { return featSize() ? static_cast<int>(initial_.size() / featSize()) : 0; }
    inline int getRecurrentInputBatchSize() const { return hiddenSize() ? static_cast<int>(rec_hidden_.size() / hiddenSize()) : 0; }

protected:
    struct Buffers {
        int B, A, HS;
        std::vector<float> policy, logit, value, reward, hidden;
        Buffers(int b, int a, int hs) : B(b), A(a), HS(hs), policy(size_t(b) * a), logit(size_t(b) * a), value(b), reward(b, 0.0f), hidden(size_t(b) * hs) {}
        std::vector<std::shared_ptr<NetworkOutput>> outputs() const
        {
            std::vector<std::shared_ptr<NetworkOutput>> outs;
            for (int i = 0; i < B; ++i) {
                auto o = std::make_shared<MuZeroNetworkOutput>(A, HS);
                std::copy(policy.begin() + size_t(i) * A, policy.begin() + size_t(i + 1) * A, o->policy_.begin());
                std::copy(logit.begin() + size_t(i) * A, logit.begin() + size_t(i + 1) * A, o->policy_logits_.begin());
                std::copy(hidden.begin() + size_t(i) * HS, hidden.begin() + size_t(i + 1) * HS, o->hidden_state_.begin());
                o->value_ = value[i];
                o->reward_ = reward[i];
                outs.emplace_back(o);
            }
            return outs;
        }
    };
    static void check(int rc)
    {
        if (rc != MZ_OK) { std::cerr << mz_last_error() << std::endl; assert(false); std::abort(); }
    }
    size_t featSize() const { return static_cast<size_t>(std::max(0, getNumInputChannels() * getInputChannelHeight() * getInputChannelWidth())); }
    int hiddenSize() const { return std::max(0, getNumHiddenChannels() * getHiddenChannelHeight() * getHiddenChannelWidth()); }
    std::mutex initial_mutex_, recurrent_mutex_;
    std::vector<float> initial_, rec_hidden_, rec_action_;
};

// ref create_network.h:11-30 (without loading the file twice)
inline std::shared_ptr<Network> createNetwork(const std::string& nn_file_name, const int gpu_id)
{
    mz_net_desc desc;
    std::vector<float> w;
    if (!readWeightFile(nn_file_name, desc, w)) { assert(false); return nullptr; }
    std::shared_ptr<Network> network;
    if (desc.type == 0) { network = std::make_shared<AlphaZeroNetwork>(); }
    else { network = std::make_shared<MuZeroNetwork>(); }
    network->loadModel(nn_file_name, gpu_id);
    return network;
}

} // namespace minizero::network
