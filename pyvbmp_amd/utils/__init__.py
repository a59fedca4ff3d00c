from .matrix_utils import *  # noqa: F401,F403
from .torch_functions import *  # noqa: F401,F403
