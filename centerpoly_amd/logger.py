"""Logger(opt) with the reference's surface (src/lib/logger.py:19-77): writes `opt.txt` and a time-stamped
`logs_*/log.txt` under opt.save_dir; `write`, `close`, `scalar_summary`.  tensorboardX is optional in
the reference and absent here: scalar summaries go to `scalars.csv` next to the log."""
import os
import sys
import time

import torch


class Logger(object):
    def __init__(self, opt):
        os.makedirs(opt.save_dir, exist_ok=True)
        os.makedirs(opt.debug_dir, exist_ok=True)
        stamp = time.strftime("%Y-%m-%d-%H-%M")
        with open(os.path.join(opt.save_dir, "opt.txt"), "wt") as f:
            f.write("==> torch version: {}\n".format(torch.__version__))
            f.write("==> hip version: {}\n".format(getattr(torch.version, "hip", None)))
            f.write("==> Cmd:\n{}\n==> Opt:\n".format(sys.argv))
            for k in sorted(n for n in dir(opt) if not n.startswith("_")):
                f.write("  %s: %s\n" % (k, getattr(opt, k)))
        self.log_dir = os.path.join(opt.save_dir, "logs_{}".format(stamp))
        os.makedirs(self.log_dir, exist_ok=True)
        self.log = open(os.path.join(self.log_dir, "log.txt"), "w")
        self._scalars = open(os.path.join(self.log_dir, "scalars.csv"), "w")
        self._scalars.write("tag,step,value\n")
        self.start_line = True

    def write(self, txt):
        if self.start_line:
            self.log.write("{}: {}".format(time.strftime("%Y-%m-%d-%H-%M"), txt))
        else:
            self.log.write(txt)
        self.start_line = "\n" in txt
        if self.start_line:
            self.log.flush()

    def scalar_summary(self, tag, value, step):
        self._scalars.write("%s,%d,%.9g\n" % (tag, step, float(value)))

    def close(self):
        self.log.close()
        self._scalars.close()
