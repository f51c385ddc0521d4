"""Helper process of the GPU test session: starts multi-rank jobs as FRESH processes.

tests/conftest.py starts this script before anything in the pytest process touches the GPU; it never imports
torch or HIP itself, so the rank processes it launches are ordinary children of a GPU-free parent (the GPU box
refuses an exec from a process that has initialised the GPU).  Protocol: one JSON object per line on stdin,
  {"argv": [...], "ranks": 2, "env": {...}, "timeout": 600}
-> the job runs once per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, and one JSON
line {"rc": [...], "tail": [...]} comes back on stdout.  EOF on stdin ends the helper.
"""
import json
import os
import socket
import subprocess
import sys
import tempfile


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        job = json.loads(line)
        n, port = int(job.get("ranks", 2)), free_port()
        procs, logs = [], []
        for r in range(n):
            env = dict(os.environ)
            env.update({k: str(v) for k, v in job.get("env", {}).items()})
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY="0")
            log = tempfile.TemporaryFile(mode="w+")
            logs.append(log)
            procs.append(subprocess.Popen(job["argv"], env=env, stdout=log, stderr=subprocess.STDOUT))
        rcs = []
        for p in procs:
            try:
                rcs.append(p.wait(timeout=float(job.get("timeout", 600))))
            except subprocess.TimeoutExpired:
                p.kill()                      # exactly the process we started
                rcs.append(-9)
        tails = []
        for log in logs:
            log.seek(0)
            tails.append(log.read()[-4000:])
            log.close()
        sys.stdout.write(json.dumps({"rc": rcs, "tail": tails}) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
