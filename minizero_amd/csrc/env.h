// Host rules engines of libmzgpu (AlphaZero leaves need game rules: ref zero_actor.cpp:55,79 replays the
// path on a copy of the root environment).  Own implementations, designed for cheap copy + replay:
// flat PODs, bitboards, early-exit flood fills, precomputed rotation tables.  Behavioural contract:
// ref environment/{tictactoe,othello,go}/*.cpp and SURVEY.md Appendix F; checked against the oracle's
// restatement by the differential playout tests.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace mz {

struct RotationTables {
    int board_size = 0;
    std::vector<int> fwd[8]; // fwd[r][a]  = getRotateAction(a, r)            (ref zero_actor.cpp:221, rotation.h:51-93)
    std::vector<int> inv[8]; // inv[r][p]  = rotate(p, reversed_rotation[r])  (ref go.cpp:289, tictactoe.cpp:77, othello.cpp:242)
    void build(int board_size, int num_actions);
};

class GameEnv {
public:
    virtual ~GameEnv() = default;
    virtual std::unique_ptr<GameEnv> clone() const = 0;
    virtual void copyFrom(const GameEnv& other) = 0; // same concrete type; cheap (no allocation in steady state)
    virtual void reset() = 0;
    // environments whose reset draws a seed from the actor's RNG stream (ref atari.h:54 reset(Random::randInt()))
    virtual bool needsSeed() const { return false; }
    virtual void resetSeed(int) { reset(); }
    virtual bool act(int action_id, int player) = 0;       // checks legality like the reference's act()
    virtual void actUnchecked(int action_id, int player) = 0; // replay of moves the search already knows are legal
    virtual bool isLegal(int action_id, int player) const = 0;
    virtual void legalMask(uint8_t* out) const = 0;         // for the player to move, all actions
    virtual bool isTerminal() const = 0;
    virtual float evalScore(bool is_resign) const = 0;
    virtual float reward() const { return 0.0f; }
    virtual void features(int rotation, float* out) const = 0;
    // the same planes, 1 bit per point (all board-game planes are 0/1): channel c occupies ceil(P/32) words, bit p%32 of word p/32.
    // 32x less host->device traffic than the f32 planes; the tower kernel expands them while staging its LDS tile
    virtual void featureBits(int rotation, uint32_t* out) const;
    int featureWords() const { return numInputChannels() * ((boardSize() * boardSize() + 31) / 32); }
    virtual int numInputChannels() const = 0;
    virtual int boardSize() const = 0;
    virtual int policySize() const = 0;
    virtual int numPlayers() const { return 2; }
    virtual std::string name() const = 0;
    virtual std::vector<std::pair<std::string, std::string>> loaderTags() const = 0;
    // ref base_env.h:216-220 (OBS tag) and atari.cpp:187-197 (L tags): only the Atari-shaped environment has either.
    // appendObservations: all kept observation strings (chw bytes per step, the initial one first) concatenated onto *out
    virtual bool hasObservations() const { return false; }
    virtual void appendObservations(std::string* /*out*/) const {}
    virtual const std::vector<int>* livesHistory() const { return nullptr; } // [i] = lives before action i
    // engines with a device twin (go_dev.hip): the root position in the device's format, once per move
    // observations that are cheaper to ship raw (bytes) and expand into float planes on the device (net_atari.hip atari_expand_features):
    // rawFeatureBytes() > 0 = supported; layout documented at the implementation
    virtual int rawFeatureBytes() const { return 0; }
    virtual void rawFeatures(uint8_t* /*dst*/) const {}
    // incremental form: the newest screen alone + the trailing (action values, valid flags) block of rawFeatures(); rawSerial() counts the screens
    // pushed so far, rawValidCount() the valid ones in the window — a consumer that saw serial - 1 (or a window with one valid screen) can rebuild
    // rawFeatures() from its previous copy shifted by one screen (worker.cpp: 1.8 MB instead of 14 MB per move over PCIe for 64 games)
    virtual uint64_t rawSerial() const { return 0; }
    virtual int rawValidCount() const { return 0; }
    virtual int rawFrameBytes() const { return 0; }
    virtual void rawNewest(uint8_t* /*frame*/, uint8_t* /*meta*/) const {}
    virtual bool hasDeviceTwin() const { return false; }
    virtual int deviceKind() const { return 0; } // GoDevView::kind (0 Go, 1 Othello)
    virtual void exportDeviceRoot(void* /*GoRootSnapshot*/) const {}
    virtual const uint64_t* zobristKeys() const { return nullptr; } // [2][points]
    virtual uint64_t turnKey() const { return 0; }                  // Go, situational superko: XORed into the hash on every move
    int turn() const { return turn_; }
    void setTurn(int player) { turn_ = player; } // BaseEnv::setTurn (ref base_env.h:103): the console's `genmove <colour>`
    // the action id of the second argument of BaseEnv::act(const std::vector<std::string>&) = {player char, action string}: board coordinates
    // ("E5", "pass"; the letter I is skipped: ref utils/sgf_loader.cpp:89-99) or an Atari action name (ref atari.cpp:9-39); -1 = not an action
    virtual int actionFromString(const std::string& s) const;
    int featureSize() const { return numInputChannels() * boardSize() * boardSize(); }
    const std::vector<int16_t>& actionIds() const { return action_ids_; }
    const std::vector<uint8_t>& actionPlayers() const { return action_players_; }
    const RotationTables* rot() const { return rot_; }

protected:
    int turn_ = 1;
    std::vector<int16_t> action_ids_;
    std::vector<uint8_t> action_players_;
    const RotationTables* rot_ = nullptr;
};

// game: "tictactoe" | "go" | "othello"; board_size 0 = the game's default (3 / 9 / 8)
// game "atari": the synthetic Atari-shaped environment (18 actions, 32 x 96 x 96 features, 1 player)
// atari_recent_observations: observation strings kept for the OBS tag (ref atari.cpp:87: intermediate sequence length + 8 + n-step +
// unrolling + 1, or everything when sequences are off)
std::unique_ptr<GameEnv> createGameEnv(const std::string& game, int board_size, float go_komi, const std::string& atari_name = "ms_pacman",
                                       int atari_episode_length = 1000, const std::string& go_ko_rule = "positional",
                                       size_t atari_recent_observations = 108001);

} // namespace mz
