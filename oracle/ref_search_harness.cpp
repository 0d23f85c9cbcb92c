// ORACLE — TEST INFRASTRUCTURE ONLY.  The REFERENCE's own self-play worker, compiled in place from /root/reference (never copied, no stand-ins): its search
// (actor/mcts.cpp, gumbel_zero.cpp), actor (zero_actor.cpp, base_actor.cpp), scheduler (actor_group.cpp), one environment (-DGO / -DOTHELLO / -DTICTACTOE:
// environment/<game>/*.cpp, environment/base/base_env.cpp), utils and config, linked against the LibTorch of the image's PyTorch.  This file is only the few
// lines of console/mode_handler.cpp:29-75,145-149 that `-mode sp` runs (the console's other modes pull in the zero server and every other game).
//   usage:  ref_search_<game> "<key=value:...>"      commands on stdin as the zero server sends them (load_model <file.pt> / start / stop / quit),
//                                                    `SelfPlay ...` records on stdout — the same protocol as apps/mzgpu_sp
// Built by `make -C oracle ref_search` ONLY where <boost/iostreams/filter/gzip.hpp> exists (utils/utils.h:4-6 needs it; this image has no Boost, so here the
// target prints why it is skipped and tests/test_ref_search.py skips).  With it, tests/test_ref_search.py pins oracle/o_mcts.cpp, o_actor.cpp and o_env.cpp
// — the "parity unpinned" half of the oracle — against the real thing.
#if !__has_include(<boost/iostreams/filter/gzip.hpp>)
#error "ref_search needs Boost (utils/utils.h includes boost/iostreams); without it the search oracle stays unpinned"
#endif
#include "actor_group.h"
#include "configuration.h"
#include "configure_loader.h"
#include "environment.h"
#include "random.h"
#include <iostream>
#include <string>

int main(int argc, char* argv[])
{
    using namespace minizero;
    env::setUpEnv();                                     // mode_handler.cpp:33
    config::ConfigureLoader cl;
    config::setConfiguration(cl);                        // mode_handler.cpp:38 (setDefaultConfiguration)
    if (argc > 1 && !cl.loadFromString(argv[1])) { std::cerr << "ref_search: bad configuration string" << std::endl; return 1; }
    utils::Random::seed(config::program_auto_seed ? static_cast<int>(time(NULL)) : config::program_seed); // mode_handler.cpp:62
    actor::ActorGroup ag;                                // mode_handler.cpp:145-149
    ag.run();
    return 0;
}
