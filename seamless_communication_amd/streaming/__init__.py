"""Streaming S2ST / S2TT (BASELINE cfg 5): the reference's SimulEval agent chain on the HIP path."""
from .agents import (DetokenizerAgent, MMATextDecoderAgent, NARUnitYUnitDecoderAgent, OfflineWav2VecBertEncoderAgent, OnlineFeatureExtractorAgent,
                     SeamlessStreamingS2STAgent, SeamlessStreamingS2TAgent, SeamlessStreamingS2TDetokAgent,
                     UnitYMMATextDecoderAgent, VocoderAgent, default_args)
from .backend import HipStreamingBackend
from .simul import EmptySegment, ReadAction, Segment, SpeechSegment, TextSegment, WriteAction

__all__ = [
    "DetokenizerAgent", "EmptySegment", "HipStreamingBackend", "MMATextDecoderAgent", "NARUnitYUnitDecoderAgent", "OfflineWav2VecBertEncoderAgent",
    "OnlineFeatureExtractorAgent", "ReadAction", "SeamlessStreamingS2STAgent", "SeamlessStreamingS2TAgent", "SeamlessStreamingS2TDetokAgent", "Segment",
    "SpeechSegment", "TextSegment", "UnitYMMATextDecoderAgent", "VocoderAgent", "WriteAction", "default_args",
]
