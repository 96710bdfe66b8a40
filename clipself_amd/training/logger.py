import logging


def setup_logging(log_file, level, include_host=False):
    fmt = logging.Formatter("%(asctime)s | %(levelname)s | %(message)s", datefmt="%Y-%m-%d,%H:%M:%S")
    logging.root.setLevel(level)
    for h in list(logging.root.handlers):
        logging.root.removeHandler(h)
    stream = logging.StreamHandler()
    stream.setFormatter(fmt)
    logging.root.addHandler(stream)
    if log_file:
        fh = logging.FileHandler(filename=log_file)
        fh.setFormatter(fmt)
        logging.root.addHandler(fh)
