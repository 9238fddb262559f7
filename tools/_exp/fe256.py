import sys; sys.argv=["kbench","none"]
sys.path.insert(0,"tools"); import kbench
kbench.frontend(128); kbench.frontend(256)
