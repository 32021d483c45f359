"""The command: lbzip2_amd/host/lbzamd.c -- lbzip2's option surface (reference src/main.c:322-627) and file
handling (:635-905) over the GPU batch path -- held side by side with the COMPILED REFERENCE program (oracle/_ref/lbzip2_stock,
the checker): same files left behind with the same bytes, modes and times, same exit status, same diagnostics (the program's
name aside), for options, environment tokens, program-name dispatch, suffix / overwrite / keep rules and damaged input.

CPU: the command linked against the emulator build of the kernels (tests/emu, small inputs).
GPU: the product binary on the MI355X -- full-size files against the reference fixtures, the reference's own decompressor
suite through `-dc` with the reference's messages (tests/golden/expand_cases.json), several devices."""
import hashlib
import json
import os
import shutil
import stat
import subprocess

import pytest

from golden_util import bench_fixtures, gen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STOCK = os.path.join(ROOT, "oracle", "_ref", "lbzip2_stock")
EMU_DIR = os.path.join(ROOT, "tests", "emu")
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def emu_cli():
    if not os.path.exists(STOCK):
        pytest.skip("oracle/_ref/lbzip2_stock (the compiled reference program) is not built")
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "WG=1024"])
    return os.path.join(EMU_DIR, "_build", "lbzamd_emu")


def _tree(d):
    """what a run left behind: name -> (permission bits, mtime in whole seconds, md5) of every file under d"""
    out = {}
    for name in sorted(os.listdir(d)):
        p = os.path.join(d, name)
        st = os.lstat(p)
        if stat.S_ISREG(st.st_mode):
            out[name] = (stat.S_IMODE(st.st_mode), int(st.st_mtime), st.st_nlink, hashlib.md5(open(p, "rb").read()).hexdigest())
        else:
            out[name] = ("dir" if stat.S_ISDIR(st.st_mode) else "other",)
    return out


def _run(prog, argv, cwd, env=None, stdin=b"", argv0=None, timeout=600):
    e = {k: v for k, v in os.environ.items() if k not in ("LBZIP2", "BZIP2", "BZIP")}
    e.update({"LBZ_EMU_THREADS": "2", "LBZAMD_POOL_SLABS": "4"})
    e.update(env or {})
    exe = prog
    if argv0:                                            # the program under another name: a link in a directory of its own
        bindir = os.path.join(os.path.dirname(cwd), "bin_" + os.path.basename(cwd))
        os.makedirs(bindir, exist_ok=True)
        exe = os.path.join(bindir, argv0)
        if not os.path.exists(exe):
            os.symlink(prog, exe)
    p = subprocess.run([exe] + argv, cwd=cwd, env=e, input=stdin, capture_output=True, timeout=timeout)
    name = os.path.basename(exe).encode()
    err = b"\n".join(l[len(name) + 2:] if l.startswith(name + b": ") else l for l in p.stderr.split(b"\n"))
    return p.returncode, p.stdout, err


def _both(tmp_path, cli, files, argv, env=None, stdin=b"", argv0=None, prepare=None, same_stdout=True, racy_words=False):
    """the same files, the same command line, each program in a directory of its own; everything observable must agree
    (racy_words: the reference's decompressor threads race on WHICH error a damaged stream is refused with -- one run says
    one thing, the next another; such streams are compared in everything but the words, and the streams whose diagnostic is
    the same in every run are pinned in tests/golden/damaged_cases.json)"""
    res = []
    for tag, prog in (("ref", STOCK), ("gpu", cli)):
        d = tmp_path / tag
        if d.exists():
            shutil.rmtree(d)
        d.mkdir()
        for name, (data, mode, mtime) in files.items():
            p = d / name
            p.write_bytes(data)
            os.chmod(p, mode)
            os.utime(p, (mtime, mtime))
        if prepare:
            prepare(d)
        rc, out, err = _run(prog, argv, str(d), env, stdin, argv0)
        res.append((rc, out if same_stdout else b"", b"" if racy_words else err, _tree(str(d))))
        last_err = err
    if racy_words:
        res[1] = res[1][:2] + (last_err,) + res[1][3:]
    assert res[0][:2] + res[0][3:] == res[1][:2] + res[1][3:] and (racy_words or res[0][2] == res[1][2]), (argv, res[0][0], res[1][0], res[0][2][-300:], res[1][2][-300:])
    return res[1]


T0 = 1_500_000_000
TEXT = bytes(gen("wiki", 120000, 2))
SOUP = bytes(gen("text", 60000, 4))
RUNS = bytes(gen("runs", 90000, 7))


def F(data, mode=0o644, mtime=T0):
    return (data, mode, mtime)


def test_compress_files_and_leave_what_lbzip2_leaves(tmp_path, emu_cli):
    # FILE -> FILE.bz2 with the input's mode and times, the input removed; two operands, one already compressed
    rc, _, err, tree = _both(tmp_path, emu_cli, {"a.txt": F(TEXT, 0o640), "b.log": F(SOUP, 0o600, T0 + 77), "c.tbz": F(b"whatever")}, ["-1", "a.txt", "b.log", "c.tbz"])
    assert rc == 4 and set(tree) == {"a.txt.bz2", "b.log.bz2", "c.tbz"} and tree["a.txt.bz2"][0] == 0o640 and tree["b.log.bz2"][1] == T0 + 77
    assert b'skipping "c.tbz": compressed suffix' in err
    # -k keeps, -v reports, a level from a cluster
    rc, _, err, tree = _both(tmp_path, emu_cli, {"a.txt": F(TEXT)}, ["-kv2", "a.txt"])
    assert rc == 0 and set(tree) == {"a.txt", "a.txt.bz2"} and b'compressing "a.txt" to "a.txt.bz2"' in err and b"compression ratio is 1:" in err
    # the output exists: skipped with a warning unless -f
    rc, _, err, tree = _both(tmp_path, emu_cli, {"a.txt": F(TEXT), "a.txt.bz2": F(b"old")}, ["-1", "a.txt"])
    assert rc == 4 and b"File exists" in err and tree["a.txt.bz2"][3] == hashlib.md5(b"old").hexdigest()
    rc, _, err, tree = _both(tmp_path, emu_cli, {"a.txt": F(TEXT), "a.txt.bz2": F(b"old")}, ["-1f", "a.txt"])
    assert rc == 0 and set(tree) == {"a.txt.bz2"}


def test_stdout_filter_and_sequential(tmp_path, emu_cli):
    rc, out, _, tree = _both(tmp_path, emu_cli, {"a.txt": F(TEXT)}, ["-1c", "a.txt"])
    assert rc == 0 and set(tree) == {"a.txt"} and out[:4] == b"BZh1"
    rc, out, _, _ = _both(tmp_path, emu_cli, {}, ["-3"], stdin=RUNS)
    assert rc == 0 and out[:4] == b"BZh3"
    rc, out, _, _ = _both(tmp_path, emu_cli, {}, ["--fast", "--sequential"], stdin=RUNS + TEXT)      # -u: blocks cut where they are full
    assert rc == 0
    rc, out, _, _ = _both(tmp_path, emu_cli, {}, ["-1"], stdin=b"")                                   # the 14-byte stream of nothing
    assert rc == 0 and len(out) == 14
    rc, _, _, tree = _both(tmp_path, emu_cli, {"empty": F(b""), "one": F(b"x")}, ["-1", "empty", "one"])        # ... as files
    assert rc == 0 and set(tree) == {"empty.bz2", "one.bz2"}
    rc, out, err, _ = _both(tmp_path, emu_cli, {}, ["-1v"], stdin=SOUP)
    assert rc == 0 and b"compressing stdin to stdout" in err and b"stdin: compression ratio" in err


def test_environment_tokens_and_program_names(tmp_path, emu_cli):
    # $LBZIP2 $BZIP2 $BZIP in front of the command line (main.c:337-354): level and -k from the environment, -v from BZIP
    rc, _, err, tree = _both(tmp_path, emu_cli, {"a.txt": F(TEXT)}, ["a.txt"], env={"LBZIP2": "-1 \t-k", "BZIP": "-v"})
    assert rc == 0 and set(tree) == {"a.txt", "a.txt.bz2"} and b"compressing" in err
    z = subprocess.run([STOCK, "-1"], input=TEXT, capture_output=True).stdout
    # bunzip2 / lbunzip2 decompress, bzcat / lbzcat decompress to stdout (main.c:376-382); -z overrides
    rc, _, _, tree = _both(tmp_path, emu_cli, {"a.txt.bz2": F(z)}, ["a.txt.bz2"], argv0="lbunzip2")
    assert rc == 0 and set(tree) == {"a.txt"} and tree["a.txt"][3] == hashlib.md5(TEXT).hexdigest()
    rc, out, _, tree = _both(tmp_path, emu_cli, {"a.txt.bz2": F(z)}, ["a.txt.bz2"], argv0="bzcat")
    assert rc == 0 and out == TEXT and set(tree) == {"a.txt.bz2"}
    rc, out, _, _ = _both(tmp_path, emu_cli, {}, ["-z1"], argv0="bunzip2", stdin=SOUP)
    assert rc == 0 and out[:4] == b"BZh1"


def test_decompress_names_test_mode_and_damage(tmp_path, emu_cli):
    z = subprocess.run([STOCK, "-1"], input=TEXT, capture_output=True).stdout
    z2 = subprocess.run([STOCK, "-2"], input=SOUP, capture_output=True).stdout
    # .bz2 stripped, .tbz2 / .tbz / .tz2 -> .tar, anything else -> .out (main.c:635-683)
    rc, _, _, tree = _both(tmp_path, emu_cli, {"a.bz2": F(z, 0o600), "b.tbz2": F(z2), "c.tz2": F(z2), "d": F(z)}, ["-d", "a.bz2", "b.tbz2", "c.tz2", "d"])
    assert rc == 0 and set(tree) == {"a", "b.tar", "c.tar", "d.out"} and tree["a"][0] == 0o600
    # -t: nothing written, nothing removed; two streams in one file; trailing garbage is ignored
    rc, out, _, tree = _both(tmp_path, emu_cli, {"a.bz2": F(z + z2 + b"\0\0trailing")}, ["-tv", "a.bz2"])
    assert rc == 0 and out == b"" and set(tree) == {"a.bz2"}
    # damaged: a flipped payload bit, a truncated file, not bzip2 at all -- lbzip2's words, status 1, no output left behind
    bad = bytearray(z); bad[len(bad) // 2] ^= 0x10
    for data in (bytes(bad), z[:len(z) * 2 // 3], z[:10], b"plain text, not bzip2\n", b""):
        rc, _, err, tree = _both(tmp_path, emu_cli, {"a.bz2": F(data)}, ["-d", "a.bz2"], racy_words=data in (bytes(bad), z[:len(z) * 2 // 3]))
        assert rc == 1 and set(tree) == {"a.bz2"} and (b"compressed data error" in err or b"not a valid bzip2 file" in err)
    # -dfc copies what is not bzip2 (process.c:675-678)
    rc, out, _, _ = _both(tmp_path, emu_cli, {}, ["-dfc"], stdin=b"plain text, not bzip2\n")
    assert rc == 0 and out == b"plain text, not bzip2\n"


def test_operands_that_are_skipped(tmp_path, emu_cli):
    def links(d):
        os.mkdir(d / "dir")
        os.link(d / "a.txt", d / "again.txt")
        os.symlink("b.txt", d / "sym.txt")
    files = {"a.txt": F(SOUP), "b.txt": F(RUNS)}
    rc, _, err, tree = _both(tmp_path, emu_cli, files, ["-1", "dir", "a.txt", "sym.txt", "missing", "b.txt"], prepare=links)
    assert rc == 4 and b"not a regular file" in err and b"more than one links" in err and b"lstat()" in err and "b.txt.bz2" in tree and "a.txt" in tree
    # -k opens files with several names; -c reads through a symbolic link
    rc, _, _, tree = _both(tmp_path, emu_cli, files, ["-1k", "a.txt"], prepare=links)
    assert rc == 0 and "a.txt.bz2" in tree
    rc, out, _, _ = _both(tmp_path, emu_cli, files, ["-1c", "sym.txt"], prepare=links)
    assert rc == 0 and out[:4] == b"BZh1"
    # "--" ends the options: a file called -x
    rc, _, _, tree = _both(tmp_path, emu_cli, {"-x": F(SOUP)}, ["-1", "--", "-x"])
    assert rc == 0 and set(tree) == {"-x.bz2"}


@pytest.mark.parametrize("argv", [["-c", "-t"], ["-tc"], ["-Q"], ["--bogus"], ["-n"], ["-n", "abc"], ["-n0"], ["-n", "2x"], ["-m", "-1"], ["-1n"]],
                         ids=lambda a: "_".join(a))
def test_option_errors_are_worded_as_lbzip2_words_them(tmp_path, emu_cli, argv):
    rc, _, err, _ = _both(tmp_path, emu_cli, {}, argv, stdin=SOUP)
    assert rc == 1 and b'specify "-h" for help' in err


def test_accepted_and_ignored_options(tmp_path, emu_cli):
    rc, out, _, _ = _both(tmp_path, emu_cli, {}, ["-1qsS", "--quiet", "--small", "--exponential", "--repetitive-best", "-n", "3", "-m2k", "-n1K"], stdin=SOUP)
    assert rc == 0 and out[:4] == b"BZh1"
    # -t then -d writes again; -d then -z compresses (main.c:288-317)
    z = subprocess.run([STOCK, "-1"], input=SOUP, capture_output=True).stdout
    rc, out, _, _ = _both(tmp_path, emu_cli, {}, ["-td"], stdin=z)
    assert rc == 0 and out == SOUP
    rc, out, _, _ = _both(tmp_path, emu_cli, {}, ["-dz1"], stdin=SOUP)
    assert rc == 0 and out == z
    for argv in (["-h"], ["--help"], ["-V"], ["--version"], ["-L"], ["-1h"]):
        rc, out, err = _run(emu_cli, argv, str(tmp_path))
        assert rc == 0 and out and not err
        assert subprocess.run([STOCK] + argv, capture_output=True).returncode == 0


def test_own_options_do_not_change_the_stream(tmp_path, emu_cli):
    data = bytes(gen("wiki", 460000, 6))
    want = subprocess.run([STOCK, "-1"], input=data, capture_output=True).stdout
    d = tmp_path / "own"
    d.mkdir()
    (d / "in").write_bytes(data)
    # file -> file with two pipelines, chunks of two slabs, several readers and writers (pread / pwrite at chunk offsets)
    rc, _, err = _run(emu_cli, ["-1k", "--pipelines=2", "--chunk-slabs=2", "--report", "-n", "3", "in"], str(d), env={"LBZ_EMU_DEVICES": "2"})
    assert rc == 0 and (d / "in.bz2").read_bytes() == want and b"file splitter/muxer" in err
    # the same through pipes (one reader, one writer, in order) and over two devices
    rc, out, _ = _run(emu_cli, ["-1", "--chunk-slabs=1", "--devices=2"], str(d), env={"LBZ_EMU_DEVICES": "2"}, stdin=data)
    assert rc == 0 and out == want
    rc, out, _ = _run(emu_cli, ["-1", "--devices=0", "--pipelines=1"], str(d), env={"LBZ_EMU_DEVICES": "3"}, stdin=data[:250001])
    assert rc == 0 and out == subprocess.run([STOCK, "-1"], input=data[:250001], capture_output=True).stdout


def test_a_failure_leaves_no_partial_output(tmp_path, emu_cli):
    z = subprocess.run([STOCK, "-1"], input=TEXT, capture_output=True).stdout
    d = tmp_path / "p"
    d.mkdir()
    (d / "a.bz2").write_bytes(z[:-7])
    rc, _, err = _run(emu_cli, ["-d", "a.bz2"], str(d))
    assert rc == 1 and sorted(os.listdir(d)) == ["a.bz2"] and b"compressed data error" in err


# ------------------------------------------------------------------ GPU: the product binary
CLI = os.path.join(ROOT, "lbzip2_amd", "host", "lbzamd")


def _cli(argv, cwd, stdin=None, env=None, timeout=600):
    p = subprocess.run([CLI] + argv, cwd=cwd, input=stdin, capture_output=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    return p.returncode, p.stdout, p.stderr


@pytest.mark.gpu
def test_gpu_cli_full_size_file_to_file(tmp_path):
    """enwik9-sized stand-in file -> file: the reference fixture's stream; back again with -d; -t; -u against its own fixture"""
    assert os.path.exists(CLI), "lbzip2_amd/host/lbzamd is not built"
    rec = [r for r in bench_fixtures() if r["kind"] == "wiki" and r["n"] == 1_000_000_000 and r["seed"] == 2][0]
    data = gen(rec["kind"], rec["n"], rec["seed"])
    d = str(tmp_path)
    with open(os.path.join(d, "enwik"), "wb") as f:
        f.write(data)
    os.chmod(os.path.join(d, "enwik"), 0o640)
    os.utime(os.path.join(d, "enwik"), (T0, T0))
    rc, _, err = _cli(["-v", "--report", "enwik"], d)
    assert rc == 0, err[-400:]
    assert sorted(os.listdir(d)) == ["enwik.bz2"]
    st = os.stat(os.path.join(d, "enwik.bz2"))
    assert st.st_size == rec["out_len"] and stat.S_IMODE(st.st_mode) == 0o640 and int(st.st_mtime) == T0
    h = hashlib.md5()
    with open(os.path.join(d, "enwik.bz2"), "rb") as f:
        for piece in iter(lambda: f.read(1 << 24), b""):
            h.update(piece)
    assert h.hexdigest() == rec["ref_md5"]
    assert b"compression ratio is 1:4.325" in err and b"file splitter/muxer" in err
    rc, _, err = _cli(["-t", "enwik.bz2"], d)
    assert rc == 0 and sorted(os.listdir(d)) == ["enwik.bz2"], err[-400:]
    rc, _, err = _cli(["-d", "enwik.bz2"], d)
    assert rc == 0 and sorted(os.listdir(d)) == ["enwik"], err[-400:]
    h = hashlib.md5()
    with open(os.path.join(d, "enwik"), "rb") as f:
        for piece in iter(lambda: f.read(1 << 24), b""):
            h.update(piece)
    assert h.hexdigest() == rec["in_md5"]
    seq = [r for r in json.load(open(os.path.join(GOLD, "seq_fixtures.json")))["records"]
           if (r["kind"], r["n"], r["seed"], r["level"]) == ("wiki", rec["n"], rec["seed"], 9)]
    if seq:
        rc, out, err = _cli(["-uc", "enwik"], d)
        assert rc == 0 and len(out) == seq[0]["out_len"] and hashlib.md5(out).hexdigest() == seq[0]["ref_md5"], err[-400:]


@pytest.mark.gpu
def test_gpu_cli_pipes_levels_and_devices(tmp_path):
    d = str(tmp_path)
    for rec in bench_fixtures(max_n=220_000_000, min_n=50_000_000):
        data = bytes(gen(rec["kind"], rec["n"], rec["seed"]))
        for extra, env in (([], None), (["--devices=2", "--pipelines=1", "--chunk-slabs=40"], {"LBZAMD_FAKE_DEVICES": "2"})):
            rc, out, err = _cli(["-%d" % rec["level"]] + extra, d, stdin=data, env=env)
            assert rc == 0 and len(out) == rec["out_len"] and hashlib.md5(out).hexdigest() == rec["canon_md5"], (rec["kind"], extra, err[-300:])
        rc, back, err = _cli(["-dc"], d, stdin=out)
        assert rc == 0 and back == data, err[-300:]


@pytest.mark.gpu
def test_gpu_cli_speaks_like_the_reference_on_its_decompressor_suite(tmp_path):
    """tests/golden/expand_cases.json: the reference's 18 decompressor cases with what the compiled reference printed and
    returned for each (`lbzip2 -dc`): same status, same bytes, same words.  damaged_cases.json: 38 hand-made and randomly
    damaged streams (test_damaged_streams_get_the_reference_s_diagnostic below)."""
    cases = json.load(open(os.path.join(GOLD, "expand_cases.json")))["cases"] + json.load(open(os.path.join(GOLD, "damaged_cases.json")))["cases"]
    for c in cases:
        rc, out, err = _cli(["-dc"], str(tmp_path), stdin=bytes.fromhex(c["bz2_hex"]))
        msg = err.decode(errors="replace").strip()
        assert rc == c["ref_exit"], (c["name"], rc, msg)
        if c["ok"]:
            assert len(out) == c["out_len"] and hashlib.md5(out).hexdigest() == c["out_md5"], c["name"]
        else:
            assert msg.replace("lbzamd:", "lbzip2_stock:") in [c["ref_message"]] + c.get("also", []), (c["name"], msg, c["ref_message"])


def test_damaged_streams_get_the_reference_s_diagnostic(emu_cli, tmp_path):
    """tests/golden/damaged_cases.json (make_damaged_fixtures.py, from the compiled reference): WHICH error a damaged stream is
    refused with.  The reference looks at things in a fixed order where one thread does the looking -- a block larger than the
    stream's level allows before its status and its CRC (expand.c:725-733), a block that ends where a run's count should stand
    before its CRC (decode.c:1009), an origin pointer behind the block / an empty block when the last code is read (:751-753),
    headers a 16-bit word at a time over input filled up to 32-bit words (parse.c:152-262, expand.c:835-842: a word that does
    not fit is a bad magic, a word that is not there the end of the file) -- and the hand-made cases pin each of these; the
    randomly damaged ones are those for which eight runs of the reference agree (for a third of such streams they do not: its
    threads race).  Same status, same words, and for the one stream that is accepted the same bytes."""
    cases = json.load(open(os.path.join(GOLD, "damaged_cases.json")))["cases"]
    assert len(cases) >= 45
    for c in cases:
        rc, out, err = _run(emu_cli, ["-dc"], str(tmp_path), stdin=bytes.fromhex(c["bz2_hex"]))[:3]
        msg = err.decode(errors="replace").strip()
        assert rc == c["ref_exit"], (c["name"], rc, msg)
        if c["ok"]:
            assert len(out) == c["out_len"] and hashlib.md5(out).hexdigest() == c["out_md5"], c["name"]
        else:                     # ("also": a defect in a block's tables or codes -- the reference's threads race between the block's own error and the parser's)
            assert msg in [m.replace("lbzip2_stock: ", "") for m in [c["ref_message"]] + c.get("also", [])], (c["name"], msg, c["ref_message"])


def test_the_bytes_in_front_of_the_damage_reach_the_pipe(emu_cli, tmp_path):
    """`-dc` of a stream whose third block is damaged: status 1, and what was written to the pipe before the diagnostic is the
    decoded data up to a block boundary -- the reference writes what it has decoded by then (whole buffers, sometimes a part of
    the damaged block's own), this command the whole blocks in front of the one it refuses.  To a FILE nothing is left behind."""
    data = bytes(gen("text", 350000, 5))
    z = bytearray(subprocess.run([STOCK, "-1"], input=data, capture_output=True).stdout)
    z[int(len(z) * 0.7)] ^= 0x20
    rc_ref, out_ref, _ = _run(STOCK, ["-dc"], str(tmp_path), stdin=bytes(z))
    rc, out, err = _run(emu_cli, ["-dc"], str(tmp_path), stdin=bytes(z))
    assert rc == rc_ref == 1 and b"compressed data error" in err
    assert data.startswith(out_ref) and data.startswith(out) and 190000 < len(out) < 210000, (len(out_ref), len(out))
    rc, _, _, tree = _both(tmp_path, emu_cli, {"a.bz2": F(bytes(z))}, ["-d", "a.bz2"], racy_words=True)
    assert rc == 1 and set(tree) == {"a.bz2"}
