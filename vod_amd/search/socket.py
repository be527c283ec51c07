"""Port helper (mirror of /root/reference/src/vod_search/socket.py:4-17)."""
import socket


def find_available_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("localhost", 0))  # port 0: the OS picks a free one
        return sock.getsockname()[1]
