"""throw-away process: can the peers' slot arrays be mapped here?  python -m jasper_amd._ipc_probe DEVICE SELF HEXHANDLES
exit code 0 = yes.  Started with a time limit by dist.shard_tables before the real attach (no torch import: starts fast)."""
import sys

from . import _lib


def main(argv):
    device, me, blob = int(argv[0]), int(argv[1]), bytes.fromhex(argv[2])
    rc = _lib.lib().jasper_ipc_probe(device, blob, len(blob) // 64, me)
    if rc:
        sys.stderr.write("ipc probe: %s\n" % _lib.lib().jasper_last_error().decode(errors="replace"))
    return 0 if rc == 0 else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
