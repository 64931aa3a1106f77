"""CPU, world_size 2 over gloo: the multi-chain plumbing (per-rank seeds, the
single gather of samples at the end, max-over-ranks timing, the rank launcher)
that the 8-GPU bench runs over RCCL.  The data path has no collective; this
covers the only exchange step."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.path.join(%(root)r, "bayes-bridge_amd"))
    os.environ["BBX_NO_TORCH"] = "0"
    import numpy as np
    import torch
    import torch.distributed as dist
    from bayesbridge_amd import chains
    rank, world, local_rank = chains.init_process_group_from_env(backend="gloo")
    assert world == 2 and dist.get_backend() == "gloo"
    seed = chains.chain_seed(111, rank)
    rng = np.random.default_rng(seed)
    local = torch.from_numpy(rng.standard_normal((5, 7)))      # [n_sample, P]
    chains.barrier()
    got = chains.gather_chain_samples(local, dst=0)
    t = chains.max_over_ranks(1.0 + rank)
    assert t == 2.0
    if rank == 0:
        assert got.shape == (2, 5, 7)
        for r in range(2):
            want = np.random.default_rng(111 + r).standard_normal((5, 7))
            assert np.array_equal(got[r].numpy(), want)
        merged = chains.merge_chain_outputs(got)
        assert merged['coef'].shape == (2, 7, 5)   # chain, coef, MCMC index last
        print("GATHER_OK")
    else:
        assert got is None
    assert chains.split_chains(5, 2, rank) == ([0, 2, 4] if rank == 0 else [1, 3])
    dist.destroy_process_group()
""")


def test_two_rank_gather_over_gloo(tmp_path):
    from bayesbridge_amd import chains
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    out = chains.launch_ranks(2, [str(script)], capture=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GATHER_OK" in out.stdout


RUN_CHAINS_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.path.join(%(root)r, "bayes-bridge_amd"))
    import numpy as np
    from bayesbridge_amd import chains

    class FakeBridge:
        # stands in for BayesBridge on a CPU-only box: gibbs() returns samples
        # that depend on the seed only
        def gibbs(self, n_iter, n_burnin, thin, seed=None, init=None,
                  params_to_save=None, coef_sampler_type=None, options=None):
            rng = np.random.default_rng(seed)
            ns = (n_iter - n_burnin) // thin
            s = {'coef': rng.standard_normal((6, ns)),
                 'global_scale': rng.random(ns), 'logp': rng.random(ns)}
            return s, {'_reg_coef_sampling_info':
                       {'n_cg_iter': np.full(ns, float(seed))}, 'seed': seed}

        # the batched route (batch='auto' / batch=2): pairs
        def batch_width(self, n_left, params_to_save=None, options=None):
            return 2 if n_left >= 2 else 0

        def gibbs_batch(self, seeds, n_iter, n_burnin, thin, init=None,
                        params_to_save=None, options=None, allow_slow=False):
            # (run_chains: an explicit width is built even where the cost
            # model refuses it, 'auto' never is)
            self.allow_slow_seen = getattr(self, 'allow_slow_seen', []) + \
                [allow_slow]
            out = []
            for slot, sd in enumerate(seeds):
                s, info = self.gibbs(n_iter, n_burnin, thin, seed=sd)
                info['batch'] = {'width': len(seeds), 'slot': slot}
                out.append((s, info))
            return out

    rank, world, _ = chains.init_process_group_from_env(backend="gloo")
    merged, infos = chains.run_chains(FakeBridge(), %(n_chain)d, 8, n_burnin=2,
                                      thin=2, seed=111)
    assert [i['chain'] for i in infos] == chains.split_chains(
        %(n_chain)d, world, rank)
    if rank == 0:
        assert merged['coef'].shape == (%(n_chain)d, 6, 3)
        for k in range(%(n_chain)d):
            rng = np.random.default_rng(111 + k)
            assert np.array_equal(merged['coef'][k],
                                  rng.standard_normal((6, 3)))
            assert np.all(merged['n_cg_iter'][k] == 111 + k)
    else:
        assert merged is None
    # default: every chain alone, and the decision is on record
    assert all(i['batch'] == {'requested': False, 'width': 1, 'slot': 0}
               for i in infos)
    # batch='auto' / an explicit width: this rank's chains in pairs, the odd
    # one alone; same seeds, same output slots
    for how in ('auto', 2):
        fake = FakeBridge()
        merged_b, infos_b = chains.run_chains(
            fake, %(n_chain)d, 8, n_burnin=2, thin=2, seed=111,
            batch=how)
        # an explicit width is the caller's decision (built even where the
        # library's cost model prices it below single chains), 'auto' is not
        assert set(getattr(fake, 'allow_slow_seen', [how != 'auto'])) \
            == {how != 'auto'}
        mine = chains.split_chains(%(n_chain)d, world, rank)
        assert [i['chain'] for i in infos_b] == mine
        widths = [2] * (len(mine) // 2 * 2) + [1] * (len(mine) %% 2)
        assert [i['batch']['width'] for i in infos_b] == widths
        assert all(i['batch']['requested'] == how for i in infos_b)
        if rank == 0:
            assert np.array_equal(merged_b['coef'], merged['coef'])
    if rank == 0:
        print("CHAINS_OK")
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()
""")


def test_run_chains_shares_chains_over_two_ranks(tmp_path):
    """config 5's plumbing: chain k gets seed 111 + k whichever rank runs it,
    odd chain counts pad, one gather per saved parameter."""
    from bayesbridge_amd import chains
    for n_chain in (4, 5):
        script = tmp_path / ("chains%d.py" % n_chain)
        script.write_text(RUN_CHAINS_WORKER % {"root": ROOT,
                                               "n_chain": n_chain})
        out = chains.launch_ranks(2, [str(script)], capture=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "CHAINS_OK" in out.stdout
    # and without a process group: all chains on this process
    script = tmp_path / "chains_single.py"
    script.write_text(RUN_CHAINS_WORKER % {"root": ROOT, "n_chain": 3})
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, str(script)], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "CHAINS_OK" in out.stdout


def test_bench_rejects_a_gpus_world_size_mismatch():
    """`--gpus` must equal the launcher's WORLD_SIZE; checked before torch or
    HIP is touched, so this runs anywhere."""
    env = dict(os.environ)
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"],
        env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 2
    assert "does not match WORLD_SIZE" in out.stderr
    assert out.stdout.strip() == ""


def test_importing_the_launcher_does_not_import_torch():
    code = ("import sys; sys.path.insert(0, %r); "
            "import bayesbridge_amd.chains; "
            "assert 'torch' not in sys.modules" %
            os.path.join(ROOT, "bayes-bridge_amd"))
    assert subprocess.run([sys.executable, "-c", code]).returncode == 0


def test_single_process_helpers_need_no_group():
    import torch
    from bayesbridge_amd import chains
    x = torch.arange(6, dtype=torch.float64).reshape(2, 3)
    got = chains.gather_chain_samples(x)
    assert got.shape == (1, 2, 3) and torch.equal(got[0], x)
    assert chains.max_over_ranks(3.5) == 3.5
    assert chains.chain_seed(111, 7) == 118      # BASELINE config 5 seeds


def test_setup_turn_serialises_processes_that_share_a_lock_file(tmp_path):
    """chains.setup_turn(): ranks that share a GPU run their device-heavy set-up
    one at a time (BBX_SETUP_LOCK; libbbx takes the same flock around its own
    device set-up).  Three processes, each holding the turn for 0.4 s: their
    intervals must not overlap.  Without the variable it is a no-op."""
    import time
    code = textwrap.dedent("""
        import os, sys, time
        sys.path.insert(0, os.path.join(%r, "bayes-bridge_amd"))
        from bayesbridge_amd import chains
        with chains.setup_turn():
            t0 = time.time()
            time.sleep(.4)
            t1 = time.time()
        print("TURN %%.3f %%.3f" %% (t0, t1))
    """ % ROOT)
    env = dict(os.environ, BBX_SETUP_LOCK=str(tmp_path / "setup.lock"))
    procs = [subprocess.Popen([sys.executable, "-c", code], env=env,
                              stdout=subprocess.PIPE, text=True)
             for _ in range(3)]
    spans = []
    for pr in procs:
        out = pr.communicate(timeout=120)[0]
        assert pr.returncode == 0
        spans.append(tuple(float(v) for v in out.split()[1:3]))
    spans.sort()
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert b0 >= a1 - 1e-3, spans
    env.pop("BBX_SETUP_LOCK")
    t = time.time()
    out = subprocess.run([sys.executable, "-c", code], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "TURN" in out.stdout


def test_setup_turn_is_reentrant_inside_a_process(tmp_path):
    """The library's lock is process-global with a depth counter: a design
    constructor (which takes it itself) may be called inside
    `with chains.setup_turn():`.  flock is per open file description -- two
    independent opens of the lock file deadlock inside ONE process, which is
    what the round-4 form did.  Nested turns must return, the lock must still
    exclude another process while held and be free afterwards; a lock path
    that cannot be opened degrades to "not serialised" with a note on stderr."""
    code = textwrap.dedent("""
        import fcntl, os, sys
        sys.path.insert(0, os.path.join(%r, "bayes-bridge_amd"))
        from bayesbridge_amd import _lib, chains
        lib = _lib.load()
        path = os.environ["BBX_SETUP_LOCK"]
        def free():
            fh = open(path, "a+")
            try:
                fcntl.flock(fh, fcntl.LOCK_EX | fcntl.LOCK_NB)
            except OSError:
                return False
            finally:
                fh.close()
            return True
        with chains.setup_turn():
            with chains.setup_turn():          # the constructor's own turn
                assert lib.bbx_setup_lock_acquire() == 1
                assert not free()
                lib.bbx_setup_lock_release()
            assert not free()                  # still held by the outer turn
        assert free()
        lib.bbx_setup_lock_release()           # unbalanced release: ignored
        assert free()
        print("NESTED_OK")
    """ % ROOT)
    env = dict(os.environ, BBX_SETUP_LOCK=str(tmp_path / "setup.lock"),
               BBX_NO_TORCH="1")
    out = subprocess.run([sys.executable, "-c", code], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "NESTED_OK" in out.stdout, out.stderr[-2000:]
    # a symlink is refused (O_NOFOLLOW) and a missing directory cannot be
    # opened: acquire returns 0 and says so once
    (tmp_path / "target").write_text("")
    os.symlink(tmp_path / "target", tmp_path / "link.lock")
    bad = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, os.path.join(%r, "bayes-bridge_amd"))
        from bayesbridge_amd import _lib, chains
        lib = _lib.load()
        assert lib.bbx_setup_lock_acquire() == 0
        with chains.setup_turn():
            pass
        print("DEGRADED_OK")
    """ % ROOT)
    for path in (tmp_path / "link.lock", tmp_path / "no" / "such" / "dir.lock"):
        env["BBX_SETUP_LOCK"] = str(path)
        out = subprocess.run([sys.executable, "-c", bad], env=env,
                             capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and "DEGRADED_OK" in out.stdout, out.stderr
        assert out.stderr.count("NOT serialised") == 1, out.stderr


def test_setup_lock_is_owned_by_a_thread(tmp_path):
    """The depth counter belongs to the THREAD that took the lock: a second
    thread of the same process waits until the holder's last release (it does
    not run its device set-up beside the holder's), and its stray release
    cannot drop the holder's flock."""
    code = textwrap.dedent("""
        import fcntl, os, sys, threading, time
        sys.path.insert(0, os.path.join(%r, "bayes-bridge_amd"))
        from bayesbridge_amd import _lib
        lib = _lib.load()
        path = os.environ["BBX_SETUP_LOCK"]
        def free():
            fh = open(path, "a+")
            try:
                fcntl.flock(fh, fcntl.LOCK_EX | fcntl.LOCK_NB)
            except OSError:
                return False
            finally:
                fh.close()
            return True
        assert lib.bbx_setup_lock_acquire() == 1       # main thread holds it
        assert lib.bbx_setup_lock_acquire() == 1       # nested: counts
        got = []
        def other():
            lib.bbx_setup_lock_release()               # not the holder: ignored
            got.append(("stray", free()))
            t0 = time.time()
            assert lib.bbx_setup_lock_acquire() == 1   # waits for the holder
            got.append(("acquired_after", time.time() - t0))
            lib.bbx_setup_lock_release()
        th = threading.Thread(target=other)
        th.start()
        time.sleep(.5)
        assert not free() and th.is_alive()            # still ours, other waits
        lib.bbx_setup_lock_release()
        time.sleep(.2)
        assert th.is_alive() and not free()            # depth 1 left
        lib.bbx_setup_lock_release()
        th.join(30)
        assert not th.is_alive()
        assert got[0] == ("stray", False), got
        assert got[1][0] == "acquired_after" and got[1][1] > .5, got
        assert free()
        print("THREAD_OWNED_OK")
    """ % ROOT)
    env = dict(os.environ, BBX_SETUP_LOCK=str(tmp_path / "setup.lock"),
               BBX_NO_TORCH="1")
    out = subprocess.run([sys.executable, "-c", code], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "THREAD_OWNED_OK" in out.stdout, \
        out.stdout[-2000:] + out.stderr[-2000:]
