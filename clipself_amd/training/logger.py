"""Process-wide log sink for `training.main`: console always, plus the experiment's out.log on the master rank.
Line layout `time | [host |] LEVEL | message` is what the reference's runs produce (src/training/logger.py:4-26), so
existing log scrapers keep working.  Re-entrant: a second call replaces the sinks instead of stacking them (the
entry point is invoked repeatedly inside one pytest process)."""
import logging
import socket

_DATE = "%Y-%m-%d,%H:%M:%S"


def _layout(include_host):
    fields = ["%(asctime)s"] + ([socket.gethostname()] if include_host else []) + ["%(levelname)s", "%(message)s"]
    return " | ".join(fields)


def setup_logging(log_file, level, include_host=False):
    sinks = [logging.StreamHandler()] + ([logging.FileHandler(log_file)] if log_file else [])
    logging.basicConfig(level=level, format=_layout(include_host), datefmt=_DATE, handlers=sinks, force=True)
    for name in list(logging.root.manager.loggerDict):       # loggers created before us follow the requested verbosity
        logging.getLogger(name).setLevel(level)
