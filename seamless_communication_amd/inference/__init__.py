from .generator import SequenceGeneratorOptions
from .translator import BatchedSpeechOutput, Modality, Task, Translator

__all__ = ["BatchedSpeechOutput", "Modality", "SequenceGeneratorOptions", "Task", "Translator"]
