from .pySim import pySim 
