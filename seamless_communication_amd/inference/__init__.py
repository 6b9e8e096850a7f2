from .generator import NGramRepeatBlockProcessor, SequenceGeneratorOptions
from .translator import BatchedSpeechOutput, Modality, Task, Translator

__all__ = ["BatchedSpeechOutput", "Modality", "NGramRepeatBlockProcessor", "SequenceGeneratorOptions", "Task", "Translator"]
