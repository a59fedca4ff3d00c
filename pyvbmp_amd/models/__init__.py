from .ARHMM import ARHMM, ARHMM_prXRY, ARHMM_prXY
from .DynamicMarkovBlanketDiscovery import DynamicMarkovBlanketDiscovery
from .GaussianMixtureModel import GaussianMixtureModel
from .HMM import HMM
from .LinearDynamicalSystems import LinearDynamicalSystems
from .MixtureofLinearDynamicalSystems import MixtureofLinearDynamicalSystems
