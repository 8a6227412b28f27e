#!/usr/bin/env python3
"""The command line's worker-process machinery (nanocall_amd/cli/nanocall.cpp: fan_out, Worker_Link, Worker_Stream, Device_Probe)
under AddressSanitizer + UndefinedBehaviorSanitizer, then under ThreadSanitizer (the parent's pump threads and the merge), on CPU.

Builds the CLI's translation unit with g++ -fsanitize=address,undefined against the shipped libnanocall_hip.so (host C++ only: no HIP
in that file) and runs the scenarios of tests/test_cli_workers_cpu.py against the instrumented binary: 1 / 2 / 5 worker processes
writing the single-process run's --stats, a worker that aborts, more workers than files.  Any sanitizer report fails the run.
(GPU sanitizers are not available on this pool; this is the CPU half, beside tools/asan_host.cpp and tools/asan_fast5.cpp.)

  python tools/asan_cli.py"""
import os
import pathlib
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def one_pass(sanitizers):
    out = tempfile.mkdtemp(prefix="asan_cli_")
    exe = os.path.join(out, "nanocall")
    lib = os.path.join(ROOT, "nanocall_amd")
    built = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + sanitizers, "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(lib, "csrc"), "-pthread", os.path.join(lib, "cli", "nanocall.cpp"), "-L" + lib, "-lnanocall_hip", "-lz",
                    "-Wl,-rpath," + lib, "-o", exe], check=False, capture_output=True, text=True)
    if built.returncode != 0:
        sys.stderr.write(built.stderr)
        raise SystemExit(built.returncode)
    import test_cli_workers_cpu as t
    t.CLI = exe
    os.environ.update(ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="report_signal_unsafe=0", NANOCALL_FULL_EXIT="1")
    plain = PLAIN_RUN[0] = PLAIN_RUN[0] or t._run

    def run(args, env=None, expect_rc=0):
        p = plain(args, env, expect_rc)
        if "Sanitizer" in p.stderr or "runtime error" in p.stderr:
            print(p.stderr[-6000:])
            raise SystemExit(1)
        return p

    t._run = run
    for w in (1, 2, 5):
        with tempfile.TemporaryDirectory() as d:
            t.test_worker_processes_write_the_stats_of_the_single_process_run(pathlib.Path(d), w)
    with tempfile.TemporaryDirectory() as d:
        t.test_a_worker_that_dies_costs_its_own_reads_only(pathlib.Path(d))
    with tempfile.TemporaryDirectory() as d:
        t.test_more_workers_than_files_and_one_file(pathlib.Path(d))
    print(f"asan_cli: -fsanitize={sanitizers}: 5 scenarios, no sanitizer report")


PLAIN_RUN = [None]


def main():
    one_pass("address,undefined")
    one_pass("thread")


if __name__ == "__main__":
    main()
