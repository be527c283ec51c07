"""Port helper (mirror of /root/reference/src/vod_search/socket.py:4-17)."""
import random
import socket


def _ephemeral_range() -> tuple[int, int]:
    try:
        with open("/proc/sys/net/ipv4/ip_local_port_range") as f:
            lo, hi = (int(v) for v in f.read().split())
        return lo, hi
    except (OSError, ValueError):
        return 32768, 60999


def find_available_port() -> int:
    """A free TCP port OUTSIDE the kernel's ephemeral range when there is one.

    The port is handed to a server that starts listening seconds later while its clients already poll it (the master's
    ping loop, the workers' rendezvous retries).  A port the OS picks for `bind(0)` lies in the ephemeral range, where
    (a) another `bind(0)` / outgoing connection may be given the same number in the meantime and (b) a client that
    keeps connecting to a not-yet-listening localhost port can be assigned that very port as its source and connect to
    itself, after which the server's `listen` fails with EADDRINUSE.  Neither can happen below the range."""
    lo, hi = _ephemeral_range()
    candidates = [p for p in range(10240, 65000) if not lo <= p <= hi]
    rng = random.SystemRandom()
    for _ in range(64):
        if not candidates:
            break
        port = rng.choice(candidates)
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
            try:
                sock.bind(("localhost", port))
            except OSError:
                continue
            return port
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("localhost", 0))  # port 0: the OS picks a free one
        return sock.getsockname()[1]


def private_socket_dir() -> str:
    """A directory for Unix-domain sockets that only the current user can write to: `$XDG_RUNTIME_DIR` when it is set and ours,
    otherwise `<tmpdir>/vodhip-<uid>` created 0700.  A socket in the shared temp dir itself could be pre-created (or removed) by
    any other user of the host; here it cannot.  Deterministic per user, so ranks that only connect derive the same path."""
    import os
    import stat
    import tempfile

    uid = os.getuid()
    xdg = os.environ.get("XDG_RUNTIME_DIR")
    if xdg:
        try:
            st = os.stat(xdg)
            if stat.S_ISDIR(st.st_mode) and st.st_uid == uid and not st.st_mode & 0o022 and os.access(xdg, os.W_OK):
                return xdg
        except OSError:
            pass
    path = os.path.join(tempfile.gettempdir(), f"vodhip-{uid}")
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != uid or st.st_mode & 0o077:
        raise PermissionError(f"{path} exists but is not a private directory of uid {uid}: refusing to place a socket there")
    return path
