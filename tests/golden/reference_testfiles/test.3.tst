gfalign evalGFA -f testFiles/random2.gfa -g testFiles/random2.gaf
embedded
+++Alignment summary+++: 
# alignments: 9
Average read length: 88.11
Average aligned sequence: 76.22
Alignment orientation (+/-): 9(100%):0(0%)
Average path length: 102.44
Average alignment quality: 60
Average matches #: 75.44
Average block length: 76.22
Primary alignments: 8
Secondary alignments: 1
Supplementary alignments: 1
Terminal supplementary alignments: 0
