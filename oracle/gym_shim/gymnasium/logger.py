import warnings


def warn(msg, *args):
    warnings.warn(msg % args if args else msg)


def info(msg, *args):
    pass


def error(msg, *args):
    warnings.warn(msg % args if args else msg)
