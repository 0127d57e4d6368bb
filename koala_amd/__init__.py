"""
koala_amd -- MI355X-native streaming noise suppressor with the Picovoice Koala surface
(`create()` / `Koala.process(frame)`), plus a batch extension for thousands of streams per GPU.
"""

from ._batch import *
from ._factory import *
from ._koala import *
from ._util import *
from .sharding import *
