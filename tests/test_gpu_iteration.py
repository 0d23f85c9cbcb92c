"""The training iteration as the zero server really drives a self-play worker, records compared with the oracle ACROSS the weight swap.

Every iteration the server writes `load_model <dir>/model/weight_iter_N.pt`, `reset_actors`, `start` (ref zero/zero_server.cpp:272-274), lets the
workers play, then `stop` (:330); before an iteration it may send `update_config <k=v:..>` (:156).  `reset_actors` is in the actors' default
ignore list (ref config/configuration.cpp:47), so by default the games in flight go on under the new weights — trees, MuZero hidden states
and Gumbel state built with the old ones.  The reference applies a command between two cycles (actor_group.cpp:200-219); the oracle's
`Group::command` does the same, and every test here sends worker and oracle the same lines at the same cycle counts — at move boundaries and
in the middle of a search — and compares every finished `SelfPlay` line and every unfinished record afterwards.

Ways in: `mz_worker_command` (the worker reads the file itself: a REAL TorchScript `.pt`, written on this box by tests/pt_writer.py — the repo's
own module with the reference's attribute names — or its `.mzw` sibling), `mz_worker_create_shared` + `mz_net_reload` (BaseActor::setNetwork),
`mz_worker_load_model` (one read for all devices), and the `-mode sp` executable with two / eight logical devices, whose devices log after how
many cycles they applied each command so that the run can be replayed on the oracle."""
import os
import re
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GO_AZ = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "alphazero")
GO_AZ_C2 = ("go_9x9", 18, 9, 9, 64, 9, 9, 1, 6, 82, 256, 1, "alphazero")      # BASELINE configs[1]'s network: sim_kernel<9,9,20,64,2>
GO_MZ = ("go_9x9", 18, 9, 9, 8, 9, 9, 1, 1, 82, 16, 1, "muzero")
GO_MZ_C4 = ("go_9x9", 18, 9, 9, 64, 9, 9, 1, 6, 82, 256, 1, "muzero")         # configs[3]'s network: sim_kernel_mz<9,9,20,68,64>
OTH_AZ = ("othello_8x8", 4, 8, 8, 8, 8, 8, 1, 1, 65, 16, 1, "alphazero")
OTH_AZ_C3 = ("othello_8x8", 4, 8, 8, 64, 8, 8, 1, 6, 65, 256, 1, "alphazero")  # configs[2]'s network: sim_kernel<8,8,4,64,0>
ATARI = ("atari_ms_pacman", 32, 96, 96, 32, 6, 6, 18, 1, 18, 32, 601, "muzero_atari")
ATARI_C5 = ("atari_ms_pacman", 32, 96, 96, 64, 6, 6, 18, 6, 18, 256, 601, "muzero_atari")  # configs[4]'s network: the Gumbel rounds' kernels
TTT = ("tictactoe", 4, 3, 3, 16, 3, 3, 1, 2, 9, 256, 1, "alphazero")

GUMBEL = ("actor_use_dirichlet_noise=false:actor_use_gumbel=true:actor_use_gumbel_noise=true:actor_gumbel_sample_size=16:"
          "actor_gumbel_sigma_visit_c=50:actor_gumbel_sigma_scale_c=1")
ATARI_CONF = ("env_game=atari:nn_type_name=muzero:actor_num_simulation=8:actor_use_dirichlet_noise=false:actor_use_gumbel=true:"
              "actor_use_gumbel_noise=true:actor_gumbel_sample_size=4:actor_gumbel_sigma_scale_c=0.1:actor_mcts_value_rescale=true:"
              "actor_mcts_reward_discount=0.997:atari_init_q=true:zero_actor_intermediate_sequence_length=6:learner_n_step_return=2:"
              "learner_muzero_unrolling_step=1:env_atari_episode_length=20:zero_num_parallel_games=4")
ATARI_C5_CONF = ("env_game=atari:nn_type_name=muzero:actor_num_simulation=50:actor_use_dirichlet_noise=false:actor_use_gumbel=true:"
                 "actor_use_gumbel_noise=true:actor_gumbel_sample_size=16:actor_gumbel_sigma_scale_c=0.1:actor_mcts_value_rescale=true:"
                 "actor_mcts_reward_discount=0.997:atari_init_q=true:zero_actor_intermediate_sequence_length=4:learner_n_step_return=1:"
                 "learner_muzero_unrolling_step=1:env_atari_episode_length=12:zero_num_parallel_games=6")
APPLY_RESET = "zero_actor_ignored_command=keep_alive"  # reset_actors out of the ignore list: the games in flight are dropped


def descs(mz, oracle, args):
    kw = dict(vh=args[10], dv=args[11], type_name=args[12])
    return mz.make_desc(*args[:10], **kw), oracle.make_desc(*args[:10], **kw)


def write_weights(mz, path, d, w, fmt):
    if fmt == "pt":
        # torch in a process of its own: importing it here would put a second copy of the HIP runtime (torch's bundled one) beside the one libmzgpu runs on
        np.save(path + ".blob.npy", np.ascontiguousarray(w, np.float32))
        open(path + ".desc.bin", "wb").write(bytes(d))
        code = ("import sys, numpy as np; sys.path[:0] = [%r, %r]; import pt_writer; from minizero_amd.lib import NetDesc; "
                "d = NetDesc.from_buffer_copy(open(%r, 'rb').read()); pt_writer.write_pt(%r, d, np.load(%r))") % (
                    os.path.join(ROOT, "tests"), ROOT, path + ".desc.bin", path, path + ".blob.npy")
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        os.remove(path + ".blob.npy")
        os.remove(path + ".desc.bin")
    else:
        from minizero_amd.export_weights import write_mzw
        write_mzw(path[:-3] + ".mzw", d, w)


class Both:
    """The worker and the oracle side by side, driven by the same protocol lines at the same cycle counts."""

    def __init__(self, mz, oracle, tmp_path, conf, args, fmt="mzw", via="worker", seed=1, threads=1):
        self.mz, self.fmt, self.via, self.dir = mz, fmt, via, str(tmp_path)
        self.d, od = descs(mz, oracle, args)
        self.games = int(re.search(r"zero_num_parallel_games=(\d+)", conf).group(1))
        w0 = mz.generate_weights(self.d, 0)
        f0 = self.file(0)
        write_weights(mz, f0, self.d, w0, fmt)
        conf = f"{conf}:program_seed={seed}:nn_file_name={f0}"
        self.og = oracle.OracleGroup(conf + ":zero_num_threads=1", od, w0)
        if via == "shared":  # BaseActor::setNetwork: the actor runs on the caller's network, the caller reloads it
            self.net = mz.Net(self.d, w0)
            self.wk = mz.Worker(conf + f":zero_num_threads={threads}", shared=self.net)
        else:                # the worker reads nn_file_name itself (ActorGroup::createNeuralNetworks)
            self.wk = mz.Worker(conf + f":zero_num_threads={threads}")
        self.cycles = 0

    def file(self, it):
        os.makedirs(os.path.join(self.dir, "model"), exist_ok=True)
        return os.path.join(self.dir, "model", f"weight_iter_{it}.pt")  # zero_server.cpp:272

    def send(self, line):
        assert self.wk.command(line) == self.og.command(line) == 0

    def load_model(self, it):
        w = self.mz.generate_weights(self.d, it)
        f = self.file(it)
        write_weights(self.mz, f, self.d, w, self.fmt)
        line = "load_model " + f
        self.og.command(line, w)
        if self.via == "shared":
            self.net.reload(w)
            self.wk.command(line)  # rename only
        elif self.via == "blob":   # the file was read by the caller, once for all devices
            d2, w2 = self.mz.read_weights_once(f)
            self.wk.load_model(f, d2, w2)
        else:
            self.wk.command(line)

    def run(self, n, expect=None):
        expect = n if expect is None else expect
        assert self.wk.run_cycles(n) == expect and self.og.cycles(n) == expect
        self.cycles += expect

    def compare(self, min_lines):
        lines, olines = self.wk.pop_lines(), self.og.lines()
        assert len(olines) >= min_lines, f"schedule too short: {len(olines)} records"
        for i, (a, b) in enumerate(zip(lines, olines)):
            assert a == b, f"line {i} differs:\n  hip   : {a[:400]}\n  oracle: {b[:400]}"
        assert len(lines) == len(olines)
        recs, orecs = self.wk.peek_records(self.games), self.og.peek_records(self.games)
        for g, (a, b) in enumerate(zip(recs, orecs)):
            assert a == b, f"unfinished record of game {g} differs:\n  hip   : {a[:400]}\n  oracle: {b[:400]}"
        assert self.wk.stats()["leaf_evals"] == self.og.leaf_evals()
        return lines, recs


def iteration(b, cpm, moves_before, mid, moves_between, moves_after, extra_conf="actor_select_action_softmax_temperature=0.5"):
    """Two server iterations + a swap in the middle of a search.  cpm = cycles per move (n + 1)."""
    b.send("start")
    b.run(cpm * moves_before)                # iteration 1 under weight_iter_0, ends AT a move boundary
    b.send("stop")
    b.run(5, expect=0)                       # stopped: nothing runs (actor_group.cpp:140)
    b.send("update_config " + extra_conf)    # zero_server.cpp:156
    b.load_model(1)                          # zero_server.cpp:272
    b.send("reset_actors")                   # :273 (ignored unless the configuration says otherwise)
    b.send("start")                          # :274
    b.run(cpm * moves_between + mid)         # iteration 2, stopped `mid` cycles INTO a search
    b.send("stop")
    b.load_model(2)
    b.send("reset_actors")
    b.send("start")
    b.run(cpm * moves_after - mid)
    b.send("keep_alive")
    b.send("stop")


CASES = {
    # name: (network, conf, cycles per move, moves before / mid-move offset / moves between / moves after, min records)
    "go_az_small": (GO_AZ, "env_game=go:env_board_size=9:actor_num_simulation=8:zero_num_parallel_games=6", 9, 30, 4, 41, 260, 6),
    "go_az_c2_net": (GO_AZ_C2, "env_game=go:env_board_size=9:actor_num_simulation=24:zero_num_parallel_games=5", 25, 11, 13, 9, 12, 0),
    "go_mz_small": (GO_MZ, "env_game=go:env_board_size=9:nn_type_name=muzero:actor_num_simulation=10:zero_num_parallel_games=4", 11, 25, 6, 33, 250, 4),
    "go_mz_c4_net": (GO_MZ_C4, "env_game=go:env_board_size=9:nn_type_name=muzero:actor_num_simulation=20:zero_num_parallel_games=4", 21, 7, 10, 8, 9, 0),
    "othello_gumbel_small": (OTH_AZ, f"env_game=othello:env_board_size=8:actor_num_simulation=16:{GUMBEL}:zero_num_parallel_games=9", 17, 20, 9, 22, 75, 9),
    "othello_gumbel_c3_net": (OTH_AZ_C3, f"env_game=othello:env_board_size=8:actor_num_simulation=16:{GUMBEL}:zero_num_parallel_games=8", 17, 9, 5, 11, 48, 8),
    "atari_gumbel_small": (ATARI, ATARI_CONF, 9, 7, 3, 9, 20, 6),
    # BASELINE configs[4]'s search and network on six games: the swap `mid` cycles into a move falls between two Gumbel rounds whose leaves are evaluated ahead
    "atari_gumbel_c5_net": (ATARI_C5, ATARI_C5_CONF, 51, 2, 21, 2, 4, 1),
    "tictactoe": (TTT, "env_game=tictactoe:actor_num_simulation=16:zero_num_parallel_games=8", 17, 12, 7, 15, 40, 40),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_iteration_with_games_in_flight(mz, oracle, tmp_path, name):
    """Default configuration: reset_actors is ignored, the games in flight continue under the new weights (hidden-state slab, Gumbel state and
    half-built trees of the old network included); one swap at a move boundary, one in the middle of a search."""
    args, conf, cpm, before, mid, between, after, min_lines = CASES[name]
    b = Both(mz, oracle, tmp_path, conf, args, threads=2)
    iteration(b, cpm, before, mid, between, after)
    lines, recs = b.compare(min_lines)
    text = "".join(lines) + "".join(recs)
    assert "EV[weight_iter_2.pt]" in text  # the tag follows the last load_model (base_actor.cpp:46)
    if name in ("go_az_c2_net", "go_mz_c4_net", "othello_gumbel_c3_net", "go_az_small", "go_mz_small", "atari_gumbel_c5_net"):
        assert b.wk.stats()["sim_launches"] > 0  # the per-game simulation kernel is what ran
    if name == "atari_gumbel_c5_net":
        assert b.wk.stats()["pre_launches"] > 0  # ... with rounds of leaves evaluated ahead


@pytest.mark.parametrize("name", ["go_az_small", "go_mz_small", "othello_gumbel_small", "atari_gumbel_small"])
def test_iteration_with_reset_actors(mz, oracle, tmp_path, name):
    """zero_actor_ignored_command without reset_actors: every actor is reset on the main thread's generator (actor_group.cpp:222-225), the games
    in flight are dropped, the next cycle is a root cycle."""
    args, conf, cpm, before, mid, between, after, min_lines = CASES[name]
    b = Both(mz, oracle, tmp_path, conf + ":" + APPLY_RESET, args, seed=4)
    iteration(b, cpm, before, mid, between, after, extra_conf="actor_resign_threshold=-0.8:actor_select_action_softmax_temperature=2")
    lines, recs = b.compare(min_lines // 2)
    assert all("EV[weight_iter_2.pt]" in r for r in recs)


@pytest.mark.parametrize("name", ["go_az_small", "go_mz_small", "atari_gumbel_small"])
def test_iteration_from_a_real_torchscript_file(mz, oracle, tmp_path, name):
    """The files are TorchScript archives written on this box (tests/pt_writer.py: the repo's own module with the reference's attribute names,
    scripted and saved exactly as learner/train.py:127 does); the worker opens and parses them itself at creation and at every load_model."""
    args, conf, cpm, before, mid, between, after, min_lines = CASES[name]
    reads = mz.weight_file_reads()
    b = Both(mz, oracle, tmp_path, conf, args, fmt="pt", seed=2)
    assert not os.path.exists(b.file(0)[:-3] + ".mzw") and open(b.file(0), "rb").read(2) == b"PK"
    iteration(b, cpm, max(3, before // 3), mid, max(3, between // 3), after)
    b.compare(min_lines)
    assert mz.weight_file_reads() - reads == 3  # creation + two load_model


@pytest.mark.parametrize("name,via", [("go_az_small", "shared"), ("go_mz_small", "shared"), ("othello_gumbel_small", "shared"), ("go_az_small", "blob"),
                                      ("atari_gumbel_small", "blob")])
def test_iteration_on_a_shared_network_or_a_blob(mz, oracle, tmp_path, name, via):
    """shared: BaseActor::setNetwork — the worker runs on the caller's mz_net (mz_worker_create_shared), the caller reloads it (mz_net_reload) and
    the worker's load_model only renames; blob: the caller has read the file (mz_weights_read) and hands the content over (mz_worker_load_model)."""
    args, conf, cpm, before, mid, between, after, min_lines = CASES[name]
    b = Both(mz, oracle, tmp_path, conf, args, via=via, seed=3)
    iteration(b, cpm, max(3, before // 2), mid, max(3, between // 2), after)
    b.compare(min_lines)


def test_load_model_refuses_another_shape(mz, oracle, tmp_path):
    b = Both(mz, oracle, tmp_path, CASES["go_az_small"][1], GO_AZ)
    d2, _ = descs(mz, oracle, ("go_9x9", 18, 9, 9, 16, 9, 9, 1, 1, 82, 16, 1, "alphazero"))
    other = os.path.join(str(tmp_path), "other.pt")
    write_weights(mz, other, d2, mz.generate_weights(d2, 0), "pt")
    with pytest.raises(mz.MzError, match="hyper-parameters"):
        b.wk.command("load_model " + other)
    with pytest.raises(mz.MzError):
        b.wk.load_model(other, d2, mz.generate_weights(d2, 0))
    with pytest.raises(mz.MzError):
        b.wk.command("load_model " + os.path.join(str(tmp_path), "missing.pt"))
    b.send("start")
    b.run(9 * 3)  # the worker still runs on weight_iter_0
    b.compare(0)


# ------------------------------------------------------------------------------------------------------------------------------------------
# through the `-mode sp` executable: the devices apply a command between two moves of their games and say after how many cycles
# ------------------------------------------------------------------------------------------------------------------------------------------
class Exe:
    def __init__(self, conf_str, game, env, exe=None):
        exe = exe or os.path.join(ROOT, "apps", "mzgpu_sp")
        assert os.path.exists(exe), "run __graft_entry__.build() first"
        self.p = subprocess.Popen([exe, "-conf_str", conf_str, "-mode", "sp", "-game", game], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                  stderr=subprocess.PIPE, text=True, env=env)
        self.lines, self.err = [], []
        self.t_out = threading.Thread(target=self._pump, args=(self.p.stdout, self.lines), daemon=True)
        self.t_err = threading.Thread(target=self._pump, args=(self.p.stderr, self.err), daemon=True)
        self.t_out.start()
        self.t_err.start()

    @staticmethod
    def _pump(stream, sink):
        for l in stream:
            sink.append(l.rstrip("\n"))

    def send(self, text):
        self.p.stdin.write(text)
        self.p.stdin.flush()

    def wait_lines(self, n, timeout=120):
        t0 = time.time()
        while len(self.lines) < n and time.time() - t0 < timeout and self.p.poll() is None:
            time.sleep(0.01)
        assert len(self.lines) >= n, f"{len(self.lines)} of {n} records after {timeout} s:\n" + "\n".join(self.err[-20:])

    def wait_err(self, pattern, count, timeout=120):
        t0 = time.time()
        while sum(1 for l in self.err if re.search(pattern, l)) < count and time.time() - t0 < timeout and self.p.poll() is None:
            time.sleep(0.01)
        assert sum(1 for l in self.err if re.search(pattern, l)) >= count, f"stderr never showed {count} x '{pattern}':\n" + "\n".join(self.err[-20:])

    def quit(self):
        self.send("quit\n")
        self.p.wait(timeout=180)
        self.t_out.join(timeout=30)
        self.t_err.join(timeout=30)
        return self.p.returncode


EXE_CASES = [
    ("tictactoe", "tictactoe", TTT, "actor_num_simulation=16", 17, 2, ""),
    ("go_reset", "go", GO_AZ, "env_board_size=9:actor_num_simulation=8", 9, 2, ":" + APPLY_RESET),
    ("othello_gumbel_eight_devices", "othello", OTH_AZ, f"env_board_size=8:actor_num_simulation=16:{GUMBEL}", 17, 8, ""),
]


def sp_executable_iterations(mz, oracle, tmp_path, game, args, extra, cpm, G, tail, exe=None, worker_extra="", make_env=None):
    """The body of test_iterations_through_the_sp_executable; `exe` / `worker_extra` / `make_env` let tests/test_sanitizers.py run the same protocol through the
    executable built over the test-only device stand-in (no GPU, ThreadSanitizer)."""
    env = dict(os.environ)
    env["MZ_DEVICE_MAP"] = ",".join(["0"] * G)
    if make_env:
        env.update(make_env)
    d, od = descs(mz, oracle, args)
    files, blobs = [], []
    os.makedirs(os.path.join(str(tmp_path), "model"))
    for it in range(3):
        blobs.append(mz.generate_weights(d, it))
        files.append(os.path.join(str(tmp_path), "model", f"weight_iter_{it}.pt"))
        write_weights(mz, files[-1], d, blobs[-1], "pt" if it != 1 else "mzw")
    games = 3 * G + 1
    conf_str = f"nn_file_name={files[0]}:program_seed=5:{extra}:zero_num_parallel_games={games}:zero_num_threads={G}{tail}{worker_extra}"
    x = Exe(conf_str, game, env, exe)
    try:
        x.send("start\n")
        x.wait_lines(2 * G)
        x.send(f"stop\nupdate_config actor_select_action_softmax_temperature=0.7\nload_model {files[1]}\nreset_actors\nstart\n")
        x.wait_err(r"device \d+: start after", 2 * G)
        n1 = len(x.lines)
        x.wait_lines(n1 + 3 * G)
        x.send(f"stop\nload_model {files[2]}\nreset_actors\nstart\n")
        x.wait_err(r"device \d+: start after", 3 * G)
        n2 = len(x.lines)
        x.wait_lines(n2 + 2 * G)
        x.send("stop\n")
        x.wait_err(r"device \d+: stop after", 3 * G)
    finally:
        rc = x.quit()
    err = "\n".join(x.err)
    assert rc == 0, err[-3000:]
    assert "Sanitizer" not in err, err[-6000:]
    assert f"{games} games on {G} GPU(s)" in err
    assert "[mzgpu] weight files read: 3" in err, err[-1500:]  # nn_file_name + two load_model, each ONCE for G devices
    # the schedule every device logged: (command, cycles)
    sched = {g: [] for g in range(G)}
    for l in x.err:
        m = re.match(r"\[mzgpu\] device (\d+): (.*) after (\d+) cycles$", l)
        if m:
            sched[int(m.group(1))].append((m.group(2), int(m.group(3))))
    expected = {}
    for g in range(G):
        assert [c.split(" ")[0] for c, _ in sched[g]] == (["start", "stop", "update_config", "load_model", "start", "stop", "load_model", "start", "stop"] if not tail else
                                                           ["start", "stop", "update_config", "load_model", "reset_actors", "start", "stop", "load_model", "reset_actors", "start", "stop"]), sched[g]
        assert all(n % cpm == 0 for _, n in sched[g]), f"device {g} applied a command inside a move: {sched[g]}"
        n_g = len(range(g, games, G))
        og = oracle.OracleGroup(f"env_game={game}:{extra}:zero_num_parallel_games={n_g}:program_seed={5 + g}:nn_file_name={files[0]}:zero_num_threads=1{tail}", od, blobs[0])
        loads = 0
        for cmd, at in sched[g]:
            if og.num_cycles() < at:
                assert og.cycles(at - og.num_cycles()) > 0
            assert og.num_cycles() == at, (cmd, at, og.num_cycles())
            if cmd.startswith("load_model"):
                loads += 1
                og.command(cmd, blobs[loads])
            else:
                og.command(cmd)
        expected[g] = og.lines()
    # every printed line is the next unseen record of exactly one device, and all of every device's records were printed (stop / quit drain them)
    cursor = {g: 0 for g in range(G)}
    for l in x.lines:
        assert l.startswith("SelfPlay ") and l.endswith(" #")
        owners = [g for g in range(G) if cursor[g] < len(expected[g]) and expected[g][cursor[g]] == l]
        assert owners, "a line that is not the next record of any device: " + l[:200]
        cursor[owners[0]] += 1
    assert all(cursor[g] == len(expected[g]) for g in range(G)), {g: (cursor[g], len(expected[g])) for g in range(G)}
    text = "".join(x.lines)
    assert "EV[weight_iter_0.pt]" in text and "EV[weight_iter_1.pt]" in text and "EV[weight_iter_2.pt]" in text


@pytest.mark.parametrize("name,game,args,extra,cpm,G,tail", EXE_CASES, ids=[c[0] for c in EXE_CASES])
def test_iterations_through_the_sp_executable(mz, oracle, tmp_path, name, game, args, extra, cpm, G, tail):
    """load_model / reset_actors / start ... stop twice through stdin with G logical devices.  Every device applies a command between two moves of
    its games (a cycle boundary the reference could have picked too) and logs after how many cycles; the test replays exactly that schedule on one
    OracleGroup per device and requires every printed record to be the next record of exactly one device — before and after the swaps — and the
    process to have read each weight file ONCE, whatever G (the reference: once per network, actor_group.cpp:227-232)."""
    sp_executable_iterations(mz, oracle, tmp_path, game, args, extra, cpm, G, tail)


# ------------------------------------------------------------------------------------------------------------------------------------------
# randomised schedules: commands at random cycle counts (inside searches, at move boundaries, back to back), random live configuration changes
# ------------------------------------------------------------------------------------------------------------------------------------------
def _seed_range(var, default):
    v = os.environ.get(var)
    if not v:
        return default
    lo, hi = v.split(":")
    return list(range(int(lo), int(hi)))


FUZZ_BASE = ["go_az_small", "go_mz_small", "othello_gumbel_small", "tictactoe", "atari_gumbel_small", "go_az_c2_net", "othello_gumbel_c3_net"]


def _live_keys(rng, gumbel, muzero):
    """configuration keys the reference reads at every use (config:: globals) and the worker does not turn into device state at creation"""
    opts = [f"actor_select_action_softmax_temperature={rng.choice([0.25, 0.5, 1.0, 2.0])}",
            f"actor_resign_threshold={rng.choice([-0.95, -0.8, -0.5, 0.3])}",
            f"zero_disable_resign_ratio={rng.choice([0.0, 0.1, 0.5, 1.0])}",
            "actor_select_action_by_count=true:actor_select_action_by_softmax_count=false",
            "actor_select_action_by_count=false:actor_select_action_by_softmax_count=true"]
    if gumbel:
        opts += [f"actor_use_gumbel_noise={rng.choice(['true', 'false'])}"]
    else:
        opts += [f"actor_dirichlet_noise_epsilon={rng.choice([0.1, 0.25, 0.5])}", f"actor_dirichlet_noise_alpha={rng.choice([0.03, 0.3, 1.0])}",
                 f"actor_use_dirichlet_noise={rng.choice(['true', 'false'])}"]
    if not muzero:
        opts += [f"actor_use_random_rotation_features={rng.choice(['true', 'false'])}"]
    k = int(rng.integers(1, 3))
    return ":".join(rng.choice(opts, size=k, replace=False))


@pytest.mark.parametrize("seed", _seed_range("MZ_FUZZ_ITER_SEEDS", list(range(14))))
def test_random_iteration_schedules(mz, oracle, tmp_path, seed):
    """A schedule drawn from the seed: which game / network / way in, whether reset_actors applies, and 3-7 segments of `run some cycles` followed by a few
    protocol lines (stop + start, update_config of live keys, load_model of other weights, reset_actors, keep_alive) — the worker and the oracle get the same lines
    at the same cycle counts; every finished line and every unfinished record must be equal at the end.  MZ_FUZZ_ITER_SEEDS=lo:hi for longer sweeps."""
    rng = np.random.default_rng(1000 + seed)
    name = FUZZ_BASE[seed % len(FUZZ_BASE)] if seed < 2 * len(FUZZ_BASE) else str(rng.choice(FUZZ_BASE))
    args, conf, cpm, *_ = CASES[name]
    via = str(rng.choice(["worker", "shared", "blob"]))
    apply_reset = rng.random() < 0.4
    b = Both(mz, oracle, tmp_path, conf + (":" + APPLY_RESET if apply_reset else ""), args, via=via, seed=int(rng.integers(1, 1000)), threads=int(rng.choice([1, 2, 4])))
    gumbel, muzero = "actor_use_gumbel=true" in conf, "nn_type_name=muzero" in conf
    heavy = name.endswith("_net") or name.startswith("atari")
    b.send("start")
    it = 0
    for _ in range(int(rng.integers(3, 8))):
        moves = int(rng.integers(1, 6 if heavy else 25))
        b.run(cpm * moves + (int(rng.integers(0, cpm)) if rng.random() < 0.7 else 0))
        stopped = rng.random() < 0.6
        if stopped:
            b.send("stop")
            if rng.random() < 0.3:
                b.run(int(rng.integers(1, 20)), expect=0)
        for _k in range(int(rng.integers(0, 4))):
            what = rng.random()
            if what < 0.35:
                b.send("update_config " + _live_keys(rng, gumbel, muzero))
            elif what < 0.7:
                it += 1
                b.load_model(it)
            elif what < 0.9:
                b.send("reset_actors")
            else:
                b.send("keep_alive")
        if stopped:
            b.send("start")
    b.run(cpm * int(rng.integers(1, 4 if heavy else 30)))
    b.compare(0)


# ------------------------------------------------------------------------------------------------------------------------------------------
# BASELINE's own pool sizes and networks, the bench's 16 RNG streams: a weight swap in the middle of a search of every game
# ------------------------------------------------------------------------------------------------------------------------------------------
FULL = {
    # key: (games, cycles before the swap (inside move 1 or 2), cycles after)
    "c2": (256, 150, 251 + 20),        # 256 games x 401 cycles: the swap after 150 simulations of move 1, then the rest of the move and 20 cycles of move 2
    "c3": (1024, 17 + 9, 8 + 17 + 3),  # 1024 games, the swap 9 simulations into move 2
    "c4": (256, 30, 21 + 51 + 4),
    "c5": (64, 51 + 23, 28 + 51 + 2),  # 64 games: the swap between two Gumbel rounds of move 2
}


@pytest.mark.slow
@pytest.mark.parametrize("key", sorted(FULL))
def test_full_size_weight_swap_inside_a_search(mz, oracle, tmp_path, key):
    from minizero_amd.export_weights import write_mzw
    games, before, after = FULL[key]
    d, od = mz.DESCS[key](), getattr(oracle, "desc_" + key)()
    head, tail = mz.CONFIGS[key].split("zero_num_parallel_games=")
    conf = head + f"zero_num_parallel_games={games}" + (":" + tail.split(":", 1)[1] if ":" in tail else "")
    files = [str(tmp_path / f"weight_iter_{i}.pt") for i in range(2)]
    blobs = [mz.generate_weights(d, i) for i in range(2)]
    for f, w in zip(files, blobs):
        write_mzw(f[:-3] + ".mzw", d, w)
    conf += f":program_seed=3:nn_file_name={files[0]}"
    og = oracle.OracleGroup(conf + ":zero_num_threads=1:oracle_throughput_threads=16", od, blobs[0])
    wk = mz.Worker(conf + f":mz_rng_streams=16:zero_num_threads={max(2, mz.usable_cpus() - 1)}")
    wk.command("start")
    assert wk.run_cycles(before) == before and og.cycles(before) == before
    for side in (wk, og):
        side.command("stop")
    assert wk.run_cycles(7) == 0 and og.cycles(7) == 0
    wk.command("update_config actor_select_action_softmax_temperature=0.8")
    og.command("update_config actor_select_action_softmax_temperature=0.8")
    wk.command("load_model " + files[1])
    og.command("load_model " + files[1], blobs[1])
    for side in (wk, og):
        side.command("reset_actors")  # ignored (the reference's default): the searches go on under the new weights
        side.command("start")
    assert wk.run_cycles(after) == after and og.cycles(after) == after
    st = wk.stats()
    assert st["leaf_evals"] == og.leaf_evals() == (before + after) * games and st["sim_launches"] > 0
    assert wk.pop_lines() == og.lines()
    recs, orecs = wk.peek_records(games), og.peek_records(games)
    for g, (a, b) in enumerate(zip(recs, orecs)):
        assert a == b, f"game {g}: records as they stand differ:\n  hip   : {a[:600]}\n  oracle: {b[:600]}"
    assert all("EV[weight_iter_1.pt]" in r for r in recs)
