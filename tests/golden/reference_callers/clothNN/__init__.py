from .controller import *
