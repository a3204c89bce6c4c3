from .tester import tester  # noqa: F401
from .trainer import trainer  # noqa: F401
from .saver import Saver  # noqa: F401
