class Error(Exception):
    pass


class DependencyNotInstalled(Error):
    pass


class NameNotFound(Error):
    pass
