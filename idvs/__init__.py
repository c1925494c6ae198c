"""Namespace package root for ``idvs.morec_amd`` (MI355X-native MoRec in-batch train step)."""
