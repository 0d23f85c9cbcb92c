// mzgpu_sp — the self-play worker executable: drop-in for `minizero_<game> -mode sp` as scripts/zero-worker.sh:160-162
// launches it:   mzgpu_sp -conf_file F -conf_str "k=v:k=v" -mode sp [-gpu N] [-game go|othello|tictactoe|atari]
// (argument parsing: ref console/mode_handler.cpp:42-57; the reference bakes the game in at compile time, here it is
//  the extra key env_game / flag -game).  Without -gpu the process drives EVERY visible GPU, like the reference's ActorGroup
//  (ref actor/actor_group.cpp:168-187): the unchanged zero-worker.sh (-g 01234567, zero_num_parallel_games = batch x #GPUs) uses the whole node.
// Build: g++ -std=c++17 -O2 apps/mzgpu_sp.cpp -Iinclude -Lminizero_amd -lmzgpu -pthread
#include "minizero/actor_group.h"
#include <fstream>

int main(int argc, char* argv[])
{
    std::string conf, conf_str, mode = "sp", game;
    int gpu = -1;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i], v = argv[i + 1];
        if (k == "-conf_file") {
            std::ifstream f(v);
            if (!f) { std::cerr << "Failed to load configure file: " << v << std::endl; return -1; }
            std::string line;
            while (std::getline(f, line)) {
                if (line.find('#') != std::string::npos) { line = line.substr(0, line.find('#')); }
                if (line.find('=') == std::string::npos) { continue; }
                conf += (conf.empty() ? "" : ":") + line;
            }
        } else if (k == "-conf_str") { conf_str = v; }
        else if (k == "-mode") { mode = v; }
        else if (k == "-gpu") { gpu = std::stoi(v); }
        else if (k == "-game") { game = v; }
        else { std::cerr << "Unknown argument: " << k << std::endl; return -1; }
    }
    if (mode != "sp") { std::cerr << "mzgpu_sp only implements -mode sp" << std::endl; return -1; }
    if (!conf_str.empty()) { conf += (conf.empty() ? "" : ":") + conf_str; }
    if (!game.empty()) { conf += ":env_game=" + game; }
    // the reference's two lines (console/mode_handler.cpp:145-149), configuration through the holder of mzgpu_config.h
    minizero::config::mzgpuConfigurationString() = conf;
    if (gpu >= 0) {
        minizero::actor::ActorGroup ag(conf, gpu);
        ag.run();
    } else {
        minizero::actor::ActorGroup ag;
        ag.run();
    }
    return 0;
}
