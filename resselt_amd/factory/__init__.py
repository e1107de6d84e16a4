from .arch import Architecture, ModelMetadata
from .key_condition import KeyCondition

__all__ = ['Architecture', 'KeyCondition', 'ModelMetadata']
