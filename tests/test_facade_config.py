"""CPU test: `actor::ActorGroup ag;` — the reference's own two lines of console/mode_handler.cpp:145-149 — compiles against the facade
next to the reference's REAL config module (config/configuration.{h,cpp}, config/configure_loader.cpp compiled where they lie), and the
facade's default constructor takes the worker's configuration from the reference's `minizero::config::*` globals.  Needs /root/reference
(skipped on the GPU box) and g++."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/minizero"

SRC = r"""
#include "configuration.h"            // the reference's
#include "minizero/actor_group.h"     // the facade
#include "minizero/actor.h"
using namespace minizero;
// console/mode_handler.cpp:145-149, verbatim
void runSelfPlay()
{
    actor::ActorGroup ag;
    ag.run();
}
int main(int argc, char** argv)
{
    config::ConfigureLoader cl;
    config::setConfiguration(cl);
    if (!cl.loadFromString(argv[1])) { return 2; }
    std::cout << config::mzgpuCollectConfiguration() << std::endl;
    if (argc > 2) { runSelfPlay(); }
    return 0;
}
"""


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree")
@pytest.mark.parametrize("game,macro", [("go", "GO"), ("othello", "OTHELLO"), ("atari", "ATARI")])
def test_actor_group_default_constructor_compiles_next_to_the_reference_config(mz, tmp_path, game, macro):
    src = tmp_path / "sp.cpp"
    src.write_text(SRC)
    exe = str(tmp_path / "sp")
    subprocess.run(["g++", "-std=c++17", "-O1", f"-D{macro}=1", str(src), os.path.join(REF, "config", "configuration.cpp"),
                    os.path.join(REF, "config", "configure_loader.cpp"), "-I" + os.path.join(REF, "config"), "-I" + os.path.join(REF, "utils"),
                    "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "minizero_amd"), "-lmzgpu", "-pthread",
                    "-Wl,-rpath," + os.path.join(ROOT, "minizero_amd"), "-o", exe], check=True, timeout=300)
    # game-specific keys are only registered for their game (ref configuration.cpp:164-196)
    conf = "actor_num_simulation=7:nn_file_name=/x/weight_iter_3.pt:zero_num_parallel_games=12:actor_use_gumbel=true:program_seed=41"
    conf += {"go": ":env_go_komi=6.5", "othello": "", "atari": ":env_atari_name=pong"}[game]
    out = subprocess.run([exe, conf], capture_output=True, text=True, timeout=60, check=True).stdout.strip()
    items = dict(kv.split("=", 1) for kv in out.split(":"))
    assert items["actor_num_simulation"] == "7" and items["nn_file_name"] == "/x/weight_iter_3.pt" and items["zero_num_parallel_games"] == "12"
    assert items["actor_use_gumbel"] == "true" and items["program_seed"] == "41" and items["env_game"] == game
    assert items["env_go_komi"] == ("6.5" if game == "go" else "7.5") and items["env_atari_name"] == ("pong" if game == "atari" else "ms_pacman")
    assert items["actor_mcts_puct_base"] == "19652" and items["zero_actor_ignored_command"] == "reset_actors"  # the reference's defaults
    # ... and the worker's own parser takes the string as it is
    e = mz.Env(out)
    assert e.policy_size() == {"go": 82, "othello": 65, "atari": 18}[game]
    # without a GPU the two lines run up to the loud "no GPU" exit of the facade (exit(0) like the reference's error paths)
    if mz.device_count() == 0:
        p = subprocess.run([exe, conf, "run"], capture_output=True, text=True, timeout=60, input="quit\n")
        assert "no GPU visible" in p.stderr
